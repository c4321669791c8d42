#!/bin/bash
# A/B of the block-scaled fp8 GEMM's tile configurations against the plain fp8 and bf16 kernels on the encoder shapes.
# Usage (GPU box): bash tools/mx_cfg_sweep.sh > gpurun_out/mx_cfg_sweep.txt
for sh in vit text; do
  echo "== bf16 (tuned) $sh"; timeout 300 python tools/gemm_bench.py --bf16 --shapes $sh 2>&1 | grep -v amdgpu.ids
  echo "== fp8 per-row scales (tuned) $sh"; timeout 300 python tools/gemm_bench.py --fp8 --shapes $sh 2>&1 | grep -v amdgpu.ids
  echo "== mx8 (tuned) $sh"; timeout 300 python tools/gemm_bench.py --mx8 --shapes $sh 2>&1 | grep -v amdgpu.ids
  for c in 0 1 2 3 4 5 6 7 8 9 10; do
    echo "== mx8 cfg $c $sh"; FERN_GEMM_MX8_CFG=$c timeout 300 python tools/gemm_bench.py --mx8 --shapes $sh 2>&1 | grep -v amdgpu.ids
  done
done
