#!/bin/bash
# Regenerate the measurement artefacts under gpurun_out/refresh on the GPU box (copy the ones to keep into profiles/):
#   bash tools/refresh_profiles.sh [round tag, default r03]          (run from the repo root, via gpurun)
# rocprofv3 needs the program itself after `--` (python3 ...), TMPDIR on /tmp, and --pmc in passes of its own; every profiled
# command runs under `timeout` (a profiler hang must not eat the GPU budget).
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
# The probe executables are git-ignored build products (they travel from the dev container with the gpurun snapshot): build any that
# is missing or older than its source -- hipcc is on the GPU box too -- and stop if one cannot be built, instead of filling the
# artefacts with "No such file" (ADVICE r3).
build_probe() {   # output, extra flags..., source is tools/probe/<first word of output's basename without _phases>.hip
    local out=$1 src=$2; shift 2
    if [ ! -x $R/$out ] || [ $R/$src -nt $R/$out ] || [ $R/fashionern_aaai2024_amd/csrc/gemm.hip -nt $R/$out ]; then
        (cd $R && hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" $src -o $out) || { echo "refresh_profiles: cannot build $out" >&2; exit 1; }
    fi
}
build_probe tools/probe/gemm_timeline tools/probe/gemm_timeline.hip -DFERN_GEMM_TRACE
build_probe tools/probe/gemm_timeline_phases tools/probe/gemm_timeline.hip -DFERN_GEMM_TRACE -DFERN_GEMM_TRACE_PHASES
build_probe tools/probe/mfma_issue_probe tools/probe/mfma_issue_probe.hip
for b in gemm_timeline gemm_timeline_phases mfma_issue_probe; do [ -x $R/tools/probe/$b ] || { echo "refresh_profiles: tools/probe/$b missing" >&2; exit 1; }; done
export FERN_HEAD=${FERN_HEAD:-$(cat $R/.fern_head 2>/dev/null || echo unknown)}      # stamped into pmc_traffic.json (no .git on the GPU box)
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
    local name=$1; shift
    rm -rf /tmp/prof_$name
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python3 $R/bench.py --no-cpu-baseline --full-record $O/${name}_under_rocprof_full.json "$@" > $O/${name}_under_rocprof.json 2> /tmp/${name}.err
    python3 $R/tools/kstats.py /tmp/prof_$name 16 $O/${name}_kernel_stats.csv > $O/${name}_kernel_stats_top.txt 2>&1
}
# tile choices of this box first; the profiled runs are pinned to them (FERN_GEMM_TILES), so their dispatch averages hold no tuner trials
rm -f /tmp/shapes.csv
FERN_PROF_DUMP=/tmp/shapes.csv timeout 600 python3 $R/bench.py --no-cpu-baseline --headline-only --lanes 1 --steps 10 --save-tiles $O/${TAG}_gemm_tiles.txt > /dev/null 2>&1
python3 $R/tools/prof_shapes.py /tmp/shapes.csv > $O/${TAG}_shapes.txt
for c in c3 c5; do      # per-shape tables of the other workloads (not pinned: their own tuner choices)
    rm -f /tmp/shapes_$c.csv
    FERN_PROF_DUMP=/tmp/shapes_$c.csv timeout 600 python3 $R/bench.py --no-cpu-baseline --headline-only --lanes 1 --steps 10 --config $c > /dev/null 2>&1
    python3 $R/tools/prof_shapes.py /tmp/shapes_$c.csv > $O/${TAG}_shapes_$c.txt
done
export FERN_GEMM_TILES=$O/${TAG}_gemm_tiles.txt
run_stats ${TAG}_bench
run_stats ${TAG}_bench_lanes1_headline_only --lanes 1 --headline-only
run_stats ${TAG}_bench_c5_lanes1 --lanes 1 --config c5
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --pmc-mode --steps 2 --lanes 1 > /tmp/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 2 $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
# the same two passes on the c5 step (block-scaled encoder + 1M-row bf16 sweep): the mx8 / bf16 GEMM families as bytes
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc5_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc5_$c -o p -- python3 $R/bench.py --pmc-mode --config c5 --steps 2 --lanes 1 > /tmp/pmc5_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py $(find /tmp/pmc5_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc5_WRITE_SIZE -name "*counter_collection.csv" | head -1) 2 $O/pmc_traffic_c5.json > $O/pmc_traffic_c5.log 2>&1
unset FERN_GEMM_TILES
timeout 300 python3 $R/tools/sweep_bench.py > $O/${TAG}_sweep_bench.txt 2>&1
rm -rf /tmp/prof_sweep
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sweep -o p -- python3 $R/tools/sweep_bench.py 1000000 > /dev/null 2>&1
python3 $R/tools/kstats.py /tmp/prof_sweep 16 $O/${TAG}_sweep_1M_kernel_stats.csv > $O/${TAG}_sweep_1M_kernel_stats_top.txt 2>&1
timeout 200 python3 $R/tools/attn_bench.py > $O/${TAG}_attn_bench.txt 2>&1
cd $R
# the driver's default command: it carries cpu_baseline, the harness / quality / PCIe legs and the c3 / c4 / c5 child runs (other_configs)
timeout 1200 python3 bench.py --full-record $O/${TAG}_bench_c2_full.json > $O/${TAG}_bench_c2.json 2> $O/${TAG}_bench_c2.err
# ranking stage: micro-benchmark of its three forms at the BASELINE shapes + kernel timelines of the forms the cost model picks
timeout 300 python3 tools/rank_bench.py > $O/${TAG}_rank_bench.txt 2>&1
cd /tmp
for v in c2:prefiltered c2:fp32_sweep c3:prefiltered 1M:prefiltered; do
    rm -rf /tmp/rtl_$v
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rtl_$v -o p -- python3 $R/tools/rank_bench.py --reps 10 --trace $v > /dev/null 2>&1
    python3 $R/tools/step_timeline.py /tmp/rtl_$v 10 --list > $O/${TAG}_rank_timeline_${v/:/_}.txt 2>&1
done
cd $R
# one-stream kernel sequence of a c2 / c5 step (what DESIGN.md's per-block attributions quote)
cd /tmp
for c in c2 c3 c5; do
    rm -rf /tmp/tl_$c
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$c -o p -- python3 $R/bench.py --pmc-mode --config $c --lanes 1 --steps 3 > /tmp/tl_$c.log 2>&1
    python3 $R/tools/step_timeline.py /tmp/tl_$c 3 --list > $O/${TAG}_timeline_$c.txt 2>&1
done
cd $R
# (the world = 8 one-GPU rehearsal of round 3 is not repeated: its artefacts are profiles/r03_bench_c*_8ranks_one_gpu_gloo.json; the multi-rank
#  paths are covered by the world-8 gloo test and the bench pre-flight)
# fp32 GEMM: per-wave timeline of the two dominant shapes + PMC view, MFMA issue-rate probe
for shp in "12608 2304 768 8 0" "12608 2304 768 12 0" "12608 768 3072 12 3"; do set -- $shp; timeout 120 tools/probe/gemm_timeline $1 $2 $3 $4 $5 /tmp/tl.csv; python3 tools/gemm_timeline.py /tmp/tl.csv; done > $O/${TAG}_gemm_timeline.txt 2>&1
timeout 60 tools/probe/mfma_issue_probe > $O/${TAG}_mfma_issue_probe.txt 2>&1
for c in 8 12; do echo "== cfg $c"; bash tools/pmc_probe.sh 12608 2304 768 $c 0; done > $O/${TAG}_pmc_gemm_probe.txt 2>&1
tail -c 400 $O/${TAG}_bench_c2.json
ls -la $O
