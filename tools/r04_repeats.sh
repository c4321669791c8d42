#!/bin/bash
# five back-to-back headline runs (c2) and five c5 runs on ONE box: run-to-run spread
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04rep
mkdir -p $O
cd $R
python - <<PY
import json, subprocess, sys
out = {}
for cfg, extra in (("c2", ["--headline-only"]), ("c5", ["--no-other-configs"])):
    rows = []
    for i in range(5):
        r = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--no-cpu-baseline", "--steps", "40"] + extra, capture_output=True, text=True, timeout=600)
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        rows.append({"value": j["value"], "ms_per_step": j["ms_per_step"], "roofline_frac": j["roofline"]["frac"], "latency_p50_ms": (j.get("latency_ms_per_batch") or {}).get("p50")})
        print(cfg, i, round(j["value"]), round(j["roofline"]["frac"], 4), flush=True)
    out[cfg] = rows
json.dump(out, open("$O/r04_bench_repeats.json", "w"), indent=1)
PY
