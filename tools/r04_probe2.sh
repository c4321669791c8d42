#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in c5 c2; do
  rm -rf /tmp/tl_$c
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$c -o p -- python3 $R/bench.py --pmc-mode --config $c --lanes 1 --steps 3 > /tmp/tl_$c.log 2>&1
  python3 $R/tools/step_timeline.py /tmp/tl_$c 3 --list > $O/timeline_$c.txt 2>&1
  head -50 $O/timeline_$c.txt
done
