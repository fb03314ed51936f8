#!/bin/bash
# cm_ffn_fused's timing-only variants INSIDE the encoder (single stream): rocprofv3 average launch duration of the two FFN kernels per
# variant of the ablation build (1 scalar GELU, 2 no GELU, 3 no weight stream after the first fill, 4 no token-fragment reads after
# the first, 5 = 2 + 3 + 4).  Results are wrong by construction; only the durations matter.   -> gpurun_out/ffn_insitu/table.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ffn_insitu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 2 3 4 5; do
  CM_LIB_PATH=$R/mamba_asr_amd/lib/libconmamba_hip_ablate.so CM_DEBUG=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$v -- \
      python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --steps 8 > $OUT/b$v.log 2>&1
  find $OUT/st$v -name "*kernel_stats.csv" -exec cp {} $OUT/stats$v.csv \;
  rm -rf $OUT/st$v
done
cd $R
python3 - <<PY > $OUT/table.txt
import csv
names = {0: "product kernel", 2: "no GELU arithmetic", 3: "no weight stream after the first fill", 4: "token fragments read once", 5: "2 + 3 + 4"}
for v in (0, 2, 3, 4, 5):
    rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("$OUT/stats%d.csv" % v))}
    f = sorted((n, t) for n, t in rows.items() if "ffn_fused_kernel" in n)
    print(f"variant {v} ({names[v]}): " + ", ".join(f"{n.split('ffn_fused_kernel')[1][:22]} {t:6.1f} us" for n, t in f))
PY
cat $OUT/table.txt
