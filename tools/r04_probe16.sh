#!/bin/bash
# f32x3 with operands split once per workgroup (configurations 7-9): bit-identity test, ViT shapes forced / tuned
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04x3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "f32x3" > $O/tests.log 2>&1
tail -3 $O/tests.log
for c in 6 7 x; do
  echo "== FERN_GEMM_SPLIT_CFG=$c"
  if [ $c = x ]; then timeout 300 python tools/gemm_bench.py --shapes vit --precision f32x3; timeout 300 python tools/gemm_bench.py --shapes big --precision f32x3
  else FERN_GEMM_SPLIT_CFG=$c timeout 300 python tools/gemm_bench.py --shapes vit --precision f32x3; FERN_GEMM_SPLIT_CFG=$c timeout 300 python tools/gemm_bench.py --shapes big --precision f32x3; fi
done > $O/gemm_bench.txt 2>&1
grep -v amdgpu.ids $O/gemm_bench.txt
