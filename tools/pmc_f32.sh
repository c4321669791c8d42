#!/bin/bash
# PMC view of the fp32 GEMM family on the encoder shapes (MFMA pipe busy, waits; LDS in a second pass): bash tools/pmc_f32.sh > gpurun_out/pmc_f32.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for sh in vit text; do
  for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
    rm -rf /tmp/pmc_f32
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_f32 -o p -- python3 $R/tools/gemm_bench.py --shapes $sh --iters 2 > /dev/null 2>&1
    echo "== fp32 $sh: $grp"
    python3 $R/tools/pmc_gemm.py $(find /tmp/pmc_f32 -name "*counter_collection.csv" | head -1)
  done
done
