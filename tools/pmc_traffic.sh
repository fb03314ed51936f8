#!/bin/bash
# Per-kernel HBM-side traffic of the forward bench (single stream, eager): separate FETCH_SIZE / WRITE_SIZE passes + the SQ pass
# of tools/pmc_layer.sh -> tools/pmc_table.py.   tools/pmc_traffic.sh <tag>   (gpurun, repository root)
set -o pipefail
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 $R/bench.py --no-cpu-baseline --streams 1 --no-graph --steps 2 --warmup 1 > $OUT/$C.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py --no-cpu-baseline --streams 1 --no-graph --steps 2 --warmup 1 > $OUT/sq.log 2>&1 || exit 1
cd $R
python3 tools/pmc_table.py $OUT/sq $OUT/FETCH_SIZE $OUT/WRITE_SIZE 14 > $OUT/table.txt
cat $OUT/table.txt
