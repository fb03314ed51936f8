#!/bin/bash
# SQ / LDS counters of every kernel of the TRAINING step (bench.py --mode train), two rocprofv3 --pmc passes (no trace domains besides
# the kernel trace):   tools/pmc_train.sh <tag> [bench.py flags]   ->  gpurun_out/pmc_<tag>/{a,b}/..., table.txt
set -o pipefail
TAG=${1:-train}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/a -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 4 --warmup 2 "$@" > $OUT/a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR \
    --kernel-trace --output-format csv -d $OUT/b -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 4 --warmup 2 "$@" > $OUT/b.log 2>&1 || exit 1
cd $R
python3 tools/pmc_sq_table.py $OUT/a $OUT/b 40 > $OUT/table.txt
rm -rf $OUT/a $OUT/b
head -c 9000 $OUT/table.txt
