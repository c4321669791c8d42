#!/bin/bash
# PMC view (MFMA pipe busy, waits, LDS activity / conflicts; separate passes per counter group) of the GEMM families on the ViT block shapes,
# tuner's plans: fp32, f32x3, block-scaled fp8 (quantising epilogue on c_fc).  bash tools/r04_pmc.sh  (via gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G2="GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
run() {   # label, gemm_bench args...
  local label=$1; shift
  for grp in "$G1" "$G2"; do
    rm -rf /tmp/pmc_x
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_x -o p -- python3 $R/tools/gemm_bench.py "$@" --iters 2 > /dev/null 2>&1
    echo "== $label: $grp"
    python3 $R/tools/pmc_gemm.py $(find /tmp/pmc_x -name "*counter_collection.csv" | head -1)
  done
}
{
run "fp32 vit (tuned plans)" --shapes vit
run "fp32 text (tuned plans)" --shapes text
run "f32x3 vit (tuned plans)" --shapes vit --precision f32x3
run "block-scaled fp8 vit (tuned, quantising c_fc)" --mx8 --mx8q --shapes vit
} > $O/r04_pmc_gemm_families.txt 2>&1
cat $O/r04_pmc_gemm_families.txt
