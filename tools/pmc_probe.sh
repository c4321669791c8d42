#!/bin/bash
# PMC view of one GEMM configuration run by the timeline probe: bash tools/pmc_probe.sh M N K cfg epi
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
  rm -rf /tmp/pmc_probe
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_probe -o p -- $R/tools/probe/gemm_timeline "$@" /tmp/pmc_probe_t.csv > /dev/null 2>&1
  python3 $R/tools/pmc_kernel.py $(find /tmp/pmc_probe -name "*counter_collection.csv" | head -1) gemm_f32
done
