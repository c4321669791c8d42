#!/bin/bash
# Kernel-level profile of the c5 workload (one lane): bash tools/prof_c5.sh   (via gpurun; output gpurun_out/c5prof/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c5prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o p -- python3 $R/bench.py --no-cpu-baseline --headline-only --lanes 1 --config c5 "$@" > $O/c5_under_rocprof.json 2> /tmp/c5.err
python3 $R/tools/kstats.py /tmp/prof_c5 30 $O/c5_kernel_stats.csv > $O/c5_kernel_stats_top.txt 2>&1
tail -3 /tmp/c5.err
cat $O/c5_kernel_stats_top.txt
