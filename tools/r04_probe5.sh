#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
echo "== fp32 exact" ; timeout 120 python tools/gemm_bench.py --shapes vit 2>&1 | grep -v amdgpu.ids
for c in 0 1 2 6; do echo "== f32x3 cfg $c"; FERN_GEMM_SPLIT_CFG=$c timeout 120 python tools/gemm_bench.py --shapes vit --precision f32x3 2>&1 | grep -v amdgpu.ids; done
echo "== f32x3 tuned"; timeout 200 python tools/gemm_bench.py --shapes vit --precision f32x3 2>&1 | grep -v amdgpu.ids
for c in 1 6; do echo "== f32x3 cfg $c big"; FERN_GEMM_SPLIT_CFG=$c timeout 120 python tools/gemm_bench.py --shapes big --precision f32x3 2>&1 | grep -v amdgpu.ids; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "f32x3" 2>&1 | tail -3
