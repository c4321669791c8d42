#!/bin/bash
# Regenerate the measurement artefacts under gpurun_out/ on the GPU box (copy the ones to keep into profiles/):
#   bash tools/refresh_profiles.sh            (run from the repo root, via gpurun)
# rocprofv3 needs the program itself after `--` (python3 ...), TMPDIR on /tmp, and --pmc in passes of its own.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
    local name=$1; shift
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/${name}_under_rocprof.json 2> /tmp/${name}.err
    python3 $R/tools/kstats.py /tmp/prof_$name 14 $O/${name}_kernel_stats.csv > $O/${name}_kernel_stats_top.txt 2>&1
}
run_stats r01_bench
run_stats r01_bench_lanes1 --lanes 1
run_stats r01_bench_lanes1_headline_only --lanes 1 --headline-only
run_stats r01_bench_bf16_lanes1 --lanes 1 --precision bf16
run_stats r01_bench_fp8_lanes1 --lanes 1 --precision fp8
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --pmc-mode --steps 2 --lanes 1 > /tmp/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 2 $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -f /tmp/shapes.csv /tmp/shapes_bf16.csv
FERN_PROF_DUMP=/tmp/shapes.csv python3 $R/bench.py --no-cpu-baseline --steps 10 > /dev/null 2>&1
python3 $R/tools/prof_shapes.py /tmp/shapes.csv > $O/r01_shapes.txt
FERN_PROF_DUMP=/tmp/shapes_bf16.csv python3 $R/bench.py --no-cpu-baseline --steps 10 --precision bf16 > /dev/null 2>&1
python3 $R/tools/prof_shapes.py /tmp/shapes_bf16.csv > $O/r01_shapes_bf16.txt
cd $R && python3 bench.py > $O/r01_bench_default.json 2> $O/r01_bench_default.err
tail -c 600 $O/r01_bench_default.json
ls -la $O
