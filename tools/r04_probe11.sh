#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py -x -q -m gpu -k "gemm or mx8 or fp8 or bf16 or fusion or resnet or batch" > $O/tests.log 2>&1
tail -3 $O/tests.log
for c in c2 c5; do
timeout 400 python bench.py --config $c --no-cpu-baseline --no-other-configs --headline-only --steps 30 > $O/$c.json 2> $O/$c.err
python - <<PY
import json
j=json.loads([l for l in open("$O/$c.json") if l.startswith("{")][-1])
r=j["roofline"]
print("$c", round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms frac", round(r["frac"],4), "gemm_ms", round(r["gemm_ms_per_step"],3))
PY
done
timeout 200 python tools/gemm_bench.py --shapes vit 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
for c in c5; do
  rm -rf /tmp/tl_$c
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$c -o p -- python3 $R/bench.py --pmc-mode --config $c --lanes 1 --steps 3 > /tmp/tl_$c.log 2>&1
  python3 $R/tools/step_timeline.py /tmp/tl_$c 3 --list > $O/timeline_$c.txt 2>&1
  head -8 $O/timeline_$c.txt; sed -n 56,64p $O/timeline_$c.txt
done
