#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_bench_contract.py -x -q -m gpu -k "gemm or bench or attention" > $O/tests.log 2>&1; tail -3 $O/tests.log
for c in c2 c5; do
timeout 400 python bench.py --config $c --no-cpu-baseline --no-other-configs --headline-only --steps 30 > $O/$c.json 2> $O/$c.err
python - <<PY
import json
j=json.loads([l for l in open("$O/$c.json") if l.startswith("{")][-1])
r=j["roofline"]
print("$c", round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms frac", round(r["frac"],4), "gemm_ms", round(r["gemm_ms_per_step"],3), "launches", r["gemm_launches_per_step"], "attn", j["attention"])
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc5; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5 -o p -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-other-configs --headline-only --lanes 1 --steps 10 > $O/c5_rocprof.json 2>/tmp/pc5.err
python3 $R/tools/kstats.py /tmp/pc5 12 | head -14
python - <<PY
import json
j=json.loads([l for l in open("$O/c5_rocprof.json") if l.startswith("{")][-1])
r=j["roofline"]; print("c5 under rocprof: frac", round(r["frac"],4), "gemm_ms", round(r["gemm_ms_per_step"],3))
PY
