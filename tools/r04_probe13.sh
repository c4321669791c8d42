#!/bin/bash
# half-wave LayerNorm of the mx8 mode: mx8 tests, c5 bench, timeline
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04ln
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py tests/test_gpu_configs.py -x -q -m gpu -s \
   -k "mx8 or c5_fp8_encoder or reduced_precision" > $O/tests.log 2>&1
grep -E "^mx8|passed|failed|Error|error" $O/tests.log | tail -15
timeout 300 python bench.py --config c5 --no-cpu-baseline --no-other-configs --steps 40 > $O/c5.json 2> $O/c5.err
python - <<PY
import json
j=json.loads([l for l in open("$O/c5.json") if l.startswith("{")][-1])
print("c5", round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms", "frac", round(j["roofline"]["frac"],4), "gemm_ms", round(j["roofline"]["gemm_ms_per_step"],3), "stage_us", j["roofline_sim_sweep"]["stage_us"], "acc", j.get("accuracy_vs_fp32_encoder"))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_c5
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_c5 -o p -- python3 $R/bench.py --pmc-mode --config c5 --lanes 1 --steps 3 > /tmp/tl_c5.log 2>&1
python3 $R/tools/step_timeline.py /tmp/tl_c5 3 --list > $O/timeline_c5.txt 2>&1
head -24 $O/timeline_c5.txt
