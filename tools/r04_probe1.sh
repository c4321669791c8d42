#!/bin/bash
# round-4 first GPU call: new parity tests, c5 lane sweep, one-stream timelines of the c5 and c2 steps
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_rank_fused.py tests/test_gpu_harness.py tests/test_gpu_configs.py -x -q -m gpu \
   -k "rank_fused or headline or goldens or in_tree or top50 or c5_fp8_encoder or exact or f32x3" > $O/tests.log 2>&1
tail -5 $O/tests.log
for lanes in 3 4 6; do
  timeout 300 python bench.py --config c5 --no-cpu-baseline --headline-only --no-other-configs --steps 40 --lanes $lanes > $O/c5_lanes$lanes.json 2> $O/c5_lanes$lanes.err
  python - <<PY
import json
j=json.loads([l for l in open("$O/c5_lanes$lanes.json") if l.startswith("{")][-1])
print("c5 lanes", $lanes, round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms", "frac", round(j["roofline"]["frac"],4), "stage_us", j["roofline_sim_sweep"]["stage_us"], "lat", j.get("latency_ms_per_batch"))
PY
done
cd /tmp && export TMPDIR=/tmp
for c in c5 c2; do
  rm -rf /tmp/tl_$c
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$c -o p -- python3 $R/bench.py --pmc-mode --config $c --lanes 1 --steps 3 > /tmp/tl_$c.log 2>&1
  python3 $R/tools/step_timeline.py /tmp/tl_$c 3 --list > $O/timeline_$c.txt 2>&1
  head -40 $O/timeline_$c.txt
done
