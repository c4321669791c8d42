#!/bin/bash
# PMC view of the attention kernels at the path's shapes (tools/attn_bench.py): matrix-pipe busy, clock, VALU / LDS issue shares, waits.
#   bash tools/pmc_attn.sh out.txt        (separate --pmc passes; the program itself after `--`)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/${1:-gpurun_out/pmc_attn.txt}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES"; do
  rm -rf /tmp/pmc_attn
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_attn -o p -- python3 $R/tools/attn_bench.py 5 > /tmp/pmc_attn.log 2>&1
  echo "## $set" >> $OUT
  python3 $R/tools/pmc_kernel.py $(find /tmp/pmc_attn -name "*counter_collection.csv" | head -1) attn >> $OUT 2>&1
done
cat $OUT
