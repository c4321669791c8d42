#!/bin/bash
# same-box A/B of pinned fp32 tile files on the headline pipeline: bash tools/c2_tiles_ab.sh base.txt variant.txt ...
run() { timeout 400 python bench.py --headline-only --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(\"{\")][-1]); print(round(d[\"value\"],1), round(d[\"ms_per_step\"],3), round(d[\"roofline\"][\"frac\"],4))"; }
for rep in 1 2; do for f in "$@"; do echo "$f"; FERN_GEMM_TILES=$f run; done; done
