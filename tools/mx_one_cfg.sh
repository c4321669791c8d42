#!/bin/bash
# time selected MX tile configurations on the ViT / text shapes: bash tools/mx_one_cfg.sh 10 7 0
for c in "$@"; do
  for sh in vit text; do
    echo "== mx8 cfg $c $sh"; FERN_GEMM_MX8_CFG=$c timeout 300 python tools/gemm_bench.py --mx8 --shapes $sh 2>&1 | grep -v amdgpu.ids
  done
done
