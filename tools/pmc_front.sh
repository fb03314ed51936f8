#!/bin/bash
# SQ counters of the front-end kernels (gpurun, from the repository root): tools/pmc_front.sh <tag>
set -o pipefail
TAG=${1:-front}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/pmc_front.py > $OUT/a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM \
    --kernel-trace --output-format csv -d $OUT/b -- python3 $R/tools/pmc_front.py > $OUT/b.log 2>&1 || exit 1
cd $R
python3 tools/pmc_counters.py $OUT/a $OUT/b --match fbank_wav > $OUT/fbank_wav.txt
python3 tools/pmc_counters.py $OUT/a $OUT/b --match cnn_front > $OUT/cnn_front.txt
cat $OUT/fbank_wav.txt $OUT/cnn_front.txt
