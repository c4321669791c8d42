#!/bin/bash
# world = 8 rehearsal on the one GPU of a dev box (gloo, ranks share the GPU): every multi-rank code path of bench.py at world 8.
# Not a performance measurement -- 8 ranks time-slice one GPU -- but shard sizes, gathers and merges are the real ones.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export FERN_BENCH_SHARE_GPU=1 FERN_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for c in c2 c3 c4 c5; do
  timeout 900 python bench.py --gpus 8 --config $c --steps 6 --warmup 3 --headline-only --no-cpu-baseline \
     > gpurun_out/r03_bench_${c}_8ranks_one_gpu_gloo.json 2> gpurun_out/r03_bench_${c}_8ranks.err
  echo "$c rc=$?"; tail -c 600 gpurun_out/r03_bench_${c}_8ranks_one_gpu_gloo.json | head -c 300; echo
done
