#!/bin/bash
# same-box A/B of the tuner's share exponent on the c5 pipeline (0 = stand-alone latency score)
run() { timeout 300 python bench.py --config c5 --headline-only --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(\"{\")][-1]); print(round(d[\"value\"]), round(d[\"ms_per_step\"],3), round(d[\"roofline\"][\"frac\"],4))"; }
for rep in 1 2; do for e in ${@:-0 0.5 1.0}; do echo "share exponent $e"; FERN_TUNE_SHARE_EXP=$e run; done; done
