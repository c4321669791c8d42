#!/bin/bash
# PMC view of the block-scaled GEMM on the ViT shapes (separate passes per counter group): bash tools/pmc_mx.sh > gpurun_out/pmc_mx.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in 0 7 4; do
  export FERN_GEMM_MX8_CFG=$cfg
  for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"; do
    rm -rf /tmp/pmc_mx
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_mx -o p -- python3 $R/tools/gemm_bench.py --mx8 --shapes vit --iters 2 > /dev/null 2>&1
    echo "== cfg $cfg: $grp"
    python3 $R/tools/pmc_gemm.py $(find /tmp/pmc_mx -name "*counter_collection.csv" | head -1)
  done
done
