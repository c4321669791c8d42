#!/bin/bash
# regenerate the per-shape tables (prof_shapes.py stopped at the stage records before)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04shapes
mkdir -p $O
cd $R
for c in c2 c3 c5; do
  rm -f /tmp/shapes_$c.csv
  FERN_PROF_DUMP=/tmp/shapes_$c.csv timeout 600 python3 bench.py --no-cpu-baseline --headline-only --lanes 1 --steps 10 --config $c > /dev/null 2>&1
  python3 tools/prof_shapes.py /tmp/shapes_$c.csv > $O/r04_shapes_$c.txt 2>&1
  wc -l $O/r04_shapes_$c.txt
done
