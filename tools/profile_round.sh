#!/bin/bash
# Collects the per-round profile set on the GPU box (run through gpurun from the repository root):
#   bench JSON line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes of the
#   dominant kernel (separate passes, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/prof_<round>/ ; copy the
#   summaries into profiles/<round>/ (tools/pmc_scan_json.py builds pmc_scan_rows.json from the two PMC passes).
set -o pipefail
ROUND=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json.log 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json.log 2>&1 || exit 1
for B in 64 32; do
  B=$B rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_b$B -- python3 $R/tools/pmc_scan.py > $OUT/pmc_fetch_b$B.log 2>&1 || exit 1
  B=$B rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_b$B -- python3 $R/tools/pmc_scan.py > $OUT/pmc_write_b$B.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/pmc_sq_b64 -- python3 $R/tools/pmc_scan.py > $OUT/pmc_sq_b64.log 2>&1 || exit 1
cd $R
python3 tools/pmc_scan_json.py $OUT > $OUT/pmc_scan_rows.json
python3 tools/pmc_counters.py $OUT/pmc_sq_b64 --match scan_rows > $OUT/pmc_scan_sq_b64.txt
tail -c 1500 $OUT/bench.json.log
