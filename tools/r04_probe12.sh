#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_rank_fused.py tests/test_gpu_kernels.py tests/test_bench_contract.py -x -q -m gpu -k "topk or rank or sweep or bench or sim" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python tools/sweep_bench.py 2>&1 | grep -v amdgpu | tee $O/sweep_bench.txt
for c in c2 c5; do
timeout 400 python bench.py --config $c --no-cpu-baseline --no-other-configs --steps 30 > $O/$c.json 2> $O/$c.err
python - <<PY
import json
j=json.loads([l for l in open("$O/$c.json") if l.startswith("{")][-1])
r=j["roofline"]
print("$c", round(j["value"]), "q/s frac", round(r["frac"],4), "rank:", {k: (round(v,1) if isinstance(v,float) else v) for k,v in j["roofline_sim_sweep"].items() if k in ("stage_us","sweep_only_us","selection_us","frac","sweep_only_GBs")})
if j.get("roofline_sim_sweep_bf16_1M"): print("  1M:", {k: (round(v,1) if isinstance(v,float) else v) for k,v in j["roofline_sim_sweep_bf16_1M"].items() if k in ("stage_us","sweep_only_us","selection_us","frac","sweep_only_GBs")})
PY
done
