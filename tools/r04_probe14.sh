#!/bin/bash
# configuration 7 of the fp32 family (16x16-tile kernel, light ring) on the text-tower / fusion shapes: forced vs tuned, and its shape suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04cfg7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tile_variant_passes and 7" > $O/tests.log 2>&1
tail -3 $O/tests.log
for c in 7 11 x; do
  echo "== FERN_GEMM_CFG=$c"
  if [ $c = x ]; then timeout 300 python tools/gemm_bench.py --shapes text; timeout 300 python tools/gemm_bench.py --shapes fusion
  else FERN_GEMM_CFG=$c timeout 300 python tools/gemm_bench.py --shapes text; FERN_GEMM_CFG=$c timeout 300 python tools/gemm_bench.py --shapes fusion; fi
done > $O/gemm_bench.txt 2>&1
cat $O/gemm_bench.txt
