#!/bin/bash
# PMC view (separate passes per counter group) of the ping-pong tile beside the 16-wave 256 x 256 tile of its family, ViT shapes + 4096^3:
#   bash tools/pmc_pp.sh > gpurun_out/pmc_pp.txt        (profiles/r06_pmc_pp.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {   # family flag, env var, cfg
  for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"; do
    for sh in vit big; do
      rm -rf /tmp/pmc_pp
      env_line="$2=$3"
      export $env_line
      timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_pp -o p -- python3 $R/tools/gemm_bench.py $1 --shapes $sh --iters 2 > /dev/null 2>&1
      unset $2
      echo "== $1 cfg $3 ($sh): $grp"
      python3 $R/tools/pmc_gemm.py $(find /tmp/pmc_pp -name "*counter_collection.csv" | head -1)
    done
  done
}
run --bf16 FERN_GEMM_BF16_CFG 2
run --bf16 FERN_GEMM_BF16_CFG 7
run --mx8 FERN_GEMM_MX8_CFG 7
run --mx8 FERN_GEMM_MX8_CFG 11
