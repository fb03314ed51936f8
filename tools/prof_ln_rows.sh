#!/bin/bash
# rocprofv3 kernel durations of cm_layernorm_bwd's kernels per A/B variant of the ablation build (tools/bench_ln_rows.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ln_rows
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 50 51; do
  CM_LIB_PATH=$R/mamba_asr_amd/lib/libconmamba_hip_ablate.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$v -- \
      python3 $R/tools/bench_ln_rows.py --debug $v > $OUT/b$v.log 2>&1
  find $OUT/st$v -name "*kernel_stats.csv" -exec cp {} $OUT/stats$v.csv \;
  rm -rf $OUT/st$v
  echo "variant $v"; grep "ln_bwd" $OUT/stats$v.csv | cut -d, -f1-4 | cut -c1-150
done
