#!/bin/bash
# fusion-tail row kernels (mean_rows in chunks with batched loads, ordered partial sums by readlane): fusion / harness tests, c2 timeline tail
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04tail
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_harness.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_c2
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_c2 -o p -- python3 $R/bench.py --pmc-mode --config c2 --lanes 1 --steps 3 > /tmp/tl_c2.log 2>&1
python3 $R/tools/step_timeline.py /tmp/tl_c2 3 --list > $O/timeline_c2.txt 2>&1
grep "mean_rows\|finalize" $O/timeline_c2.txt | head
cd $R
timeout 300 python bench.py --no-cpu-baseline --headline-only --steps 30 > $O/c2.json 2>/dev/null
python - <<PY
import json
j=json.loads([l for l in open("$O/c2.json") if l.startswith("{")][-1])
print("c2", round(j["value"]), round(j["roofline"]["frac"],4))
PY
