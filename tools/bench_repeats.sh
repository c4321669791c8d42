#!/bin/bash
# run-to-run spread on ONE box: five back-to-back headline-only runs of c2 (fp32) and of c5 (mx8img).   bash tools/bench_repeats.sh out.json
OUT=${1:-gpurun_out/bench_repeats.json}
python - "$OUT" <<'PY'
import json, subprocess, sys
out = {}
for cfg in ("c2", "c5"):
    runs = []
    for _ in range(5):
        r = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--headline-only", "--no-cpu-baseline", "--no-other-configs", "--steps", "40",
                            "--full-record", "/tmp/fern_repeat_full.json"], capture_output=True, text=True, timeout=600)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        runs.append({"value": d["value"], "ms_per_step": d["ms_per_step"], "roofline_frac": d["roofline"]["frac"], "precision": d["encoder_precision"]})
    v = [x["value"] for x in runs]
    out[cfg] = {"runs": runs, "min": min(v), "max": max(v), "spread_pct": (max(v) - min(v)) / (sum(v) / len(v)) * 100}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: (round(v["min"]), round(v["max"]), round(v["spread_pct"], 2)) for k, v in out.items()}))
PY
