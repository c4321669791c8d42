#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention or paired" > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 300 python tools/attn_bench.py > $O/attn_pair.txt 2>&1
FERN_ATTN_PAIR=0 timeout 300 python tools/attn_bench.py > $O/attn_nopair.txt 2>&1
echo "== paired"; cat $O/attn_pair.txt; echo "== one head per workgroup"; cat $O/attn_nopair.txt
