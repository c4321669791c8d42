#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py -x -q -m gpu -k "mx8" > $O/tests.log 2>&1
tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_c5
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_c5 -o p -- python3 $R/bench.py --pmc-mode --config c5 --lanes 1 --steps 3 > /tmp/tl_c5.log 2>&1
python3 $R/tools/step_timeline.py /tmp/tl_c5 3 --list > $O/timeline_c5.txt 2>&1
head -12 $O/timeline_c5.txt; sed -n 58,72p $O/timeline_c5.txt
cd $R
timeout 300 python bench.py --config c5 --no-cpu-baseline --no-other-configs --headline-only --steps 40 > $O/c5.json 2> $O/c5.err
python - <<PY
import json
j=json.loads([l for l in open("$O/c5.json") if l.startswith("{")][-1])
print("c5", round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms", "frac", round(j["roofline"]["frac"],4), "gemm_ms", round(j["roofline"]["gemm_ms_per_step"],3))
PY
