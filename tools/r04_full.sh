#!/bin/bash
# full GPU suite + the driver's default bench command
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04full
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
tail -4 $O/tests.log
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
j=json.loads([l for l in open("$O/bench_c2.json") if l.startswith("{")][-1])
print("c2", round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms frac", round(j["roofline"]["frac"],4), "step_level", round(j["roofline"]["step_level"]["frac"],4), "lat", j["latency_ms_per_batch"]["p50"], j["latency_ms_per_batch"]["p99"])
print("rank", j["roofline_sim_sweep"]["stage_us"], "1M", j["roofline_sim_sweep_bf16_1M"]["stage_us"], j["roofline_sim_sweep_bf16_1M"]["frac"])
for k in ("encoder_f32x3","encoder_bf16","encoder_fp8","encoder_mx8"):
    print(k, round(j[k]["value"]), round(j[k]["gemm_frac"],4), j[k]["vs_fp32_top50_overlap"])
print("lookup", j["lookup_variant"]["value"])
for k,v in j["other_configs"].items():
    print(k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("value","ms_per_step","rank_stage_us","rank_stage_frac_of_hbm")}, v.get("roofline",{}).get("frac"), v.get("error"))
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["parity_vs_hip"])
PY
