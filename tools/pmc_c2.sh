#!/bin/bash
# PMC traffic passes of the c2 step alone (superseded by tools/final_check.sh, which also pins the tuner choices of both passes)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
export FERN_HEAD=$(cat $R/.fern_head 2>/dev/null || echo unknown)
export FERN_GEMM_TILES=$O/r05_gemm_tiles.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --pmc-mode --steps 2 --lanes 1 > /tmp/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 2 $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
grep -A8 '"sweep"' $O/pmc_traffic.json
