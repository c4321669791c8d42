#!/bin/bash
# one-stream timeline of a c3 step (RN50x4 image tower)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_c3
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_c3 -o p -- python3 $R/bench.py --pmc-mode --config c3 --lanes 1 --steps 3 > /tmp/tl_c3.log 2>&1
python3 $R/tools/step_timeline.py /tmp/tl_c3 3 --list > $O/timeline_c3.txt 2>&1
head -40 $O/timeline_c3.txt
FERN_PROF_DUMP=/tmp/shapes_c3.csv timeout 600 python3 $R/bench.py --no-cpu-baseline --headline-only --lanes 1 --steps 6 --config c3 > /dev/null 2>&1
python3 $R/tools/prof_shapes.py /tmp/shapes_c3.csv > $O/shapes_c3.txt; head -50 $O/shapes_c3.txt
