#!/bin/bash
# Regenerate the measurement artefacts under gpurun_out/refresh on the GPU box (copy the ones to keep into profiles/):
#   bash tools/refresh_profiles.sh [round tag, default r02]          (run from the repo root, via gpurun)
# rocprofv3 needs the program itself after `--` (python3 ...), TMPDIR on /tmp, and --pmc in passes of its own; every profiled
# command runs under `timeout` (a profiler hang must not eat the GPU budget).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
    local name=$1; shift
    rm -rf /tmp/prof_$name
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/${name}_under_rocprof.json 2> /tmp/${name}.err
    python3 $R/tools/kstats.py /tmp/prof_$name 16 $O/${name}_kernel_stats.csv > $O/${name}_kernel_stats_top.txt 2>&1
}
# tile choices of this box first; the profiled runs are pinned to them (FERN_GEMM_TILES), so their dispatch averages hold no tuner trials
rm -f /tmp/shapes.csv
FERN_PROF_DUMP=/tmp/shapes.csv timeout 600 python3 $R/bench.py --no-cpu-baseline --headline-only --lanes 1 --steps 10 --save-tiles $O/${TAG}_gemm_tiles.txt > /dev/null 2>&1
python3 $R/tools/prof_shapes.py /tmp/shapes.csv > $O/${TAG}_shapes.txt
export FERN_GEMM_TILES=$O/${TAG}_gemm_tiles.txt
run_stats ${TAG}_bench
run_stats ${TAG}_bench_lanes1_headline_only --lanes 1 --headline-only
run_stats ${TAG}_bench_c5_lanes1 --lanes 1 --config c5
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --pmc-mode --steps 2 --lanes 1 > /tmp/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 2 $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
unset FERN_GEMM_TILES
timeout 300 python3 $R/tools/sweep_bench.py > $O/${TAG}_sweep_bench.txt 2>&1
rm -rf /tmp/prof_sweep
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sweep -o p -- python3 $R/tools/sweep_bench.py 1000000 > /dev/null 2>&1
python3 $R/tools/kstats.py /tmp/prof_sweep 16 $O/${TAG}_sweep_1M_kernel_stats.csv > $O/${TAG}_sweep_1M_kernel_stats_top.txt 2>&1
timeout 200 python3 $R/tools/attn_bench.py > $O/${TAG}_attn_bench.txt 2>&1
cd $R
for c in c2 c3 c4 c5; do
    timeout 900 python3 bench.py --config $c $( [ $c = c2 ] || echo --no-cpu-baseline ) > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err
done
FERN_DIST_BACKEND=gloo FERN_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --headline-only > $O/${TAG}_bench_2ranks_one_gpu_gloo.json 2> $O/${TAG}_bench_2ranks_one_gpu_gloo.err
FERN_DIST_BACKEND=gloo FERN_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --config c5 --headline-only > $O/${TAG}_bench_c5_2ranks_one_gpu_gloo.json 2> $O/${TAG}_bench_c5_2ranks_one_gpu_gloo.err
timeout 600 bash tools/mx_cfg_sweep.sh > $O/${TAG}_mx_cfg_sweep.txt 2>&1
tail -c 400 $O/${TAG}_bench_c2.json
ls -la $O
