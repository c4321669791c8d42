#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py tests/test_gpu_harness.py -x -q -m gpu -k "layernorm or mx8 or goldens or in_tree or batch_invar or towers or f32x3" > $O/tests.log 2>&1
tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for c in c5 c2; do
  rm -rf /tmp/tl_$c
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$c -o p -- python3 $R/bench.py --pmc-mode --config $c --lanes 1 --steps 3 > /tmp/tl_$c.log 2>&1
  python3 $R/tools/step_timeline.py /tmp/tl_$c 3 --list > $O/timeline_$c.txt 2>&1
  head -3 $O/timeline_$c.txt; grep -i "layernorm" $O/timeline_$c.txt | head -4
done
cd $R
timeout 300 python bench.py --config c5 --no-cpu-baseline --no-other-configs --headline-only --steps 40 > $O/c5.json 2> $O/c5.err
python - <<PY
import json
j=json.loads([l for l in open("$O/c5.json") if l.startswith("{")][-1])
print("c5", round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms", "frac", round(j["roofline"]["frac"],4))
PY
