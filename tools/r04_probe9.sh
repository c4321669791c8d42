#!/bin/bash
# block-scaled GEMM: does a start-up stagger of the second workgroup of every CU de-phase the co-resident pairs?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in 9 0 5; do
  for st in 0 3 6 9; do
    echo "== cfg $cfg stagger $st us"
    if [ $st = 0 ]; then FERN_GEMM_MX8_CFG=$cfg timeout 120 python tools/gemm_bench.py --shapes vit --mx8 --mx8q --iters 50 2>&1 | grep -v amdgpu.ids
    else FERN_GEMM_MX8_CFG=$cfg FERN_MX8_STAGGER=$st timeout 120 python tools/gemm_bench.py --shapes vit --mx8 --mx8q --iters 50 2>&1 | grep -v amdgpu.ids; fi
  done
done
