#!/bin/bash
# PMC passes over tools/pmc_ffn.py (cm_ffn_fused alone, 64k rows): persistent kernel (default) and the one-tile-per-workgroup
# kernel (CM_DEBUG=20).  Output: gpurun_out/pmc_ffn_<tag>_<pass>/ ; summarise with tools/pmc_counters.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in persist legacy; do
  if [ $tag = legacy ]; then export CM_DEBUG=20; else unset CM_DEBUG; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ffn_${tag}_a -- python3 $R/tools/pmc_ffn.py > $R/gpurun_out/pmc_ffn_${tag}_a.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM \
      --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ffn_${tag}_b -- python3 $R/tools/pmc_ffn.py > $R/gpurun_out/pmc_ffn_${tag}_b.log 2>&1 || echo "pass b failed ($tag)"
done
