#!/bin/bash
# same-box A/B of pinned tile files on the c5 pipeline: bash tools/c5_tiles_ab.sh file...   (each alternated with the tuner's own choices)
run() { timeout 300 python bench.py --config c5 --headline-only --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(\"{\")][-1]); print(round(d[\"value\"]), round(d[\"ms_per_step\"],3), round(d[\"roofline\"][\"frac\"],4))"; }
for rep in 1 2; do
  echo "tuned"; run
  for f in "$@"; do echo "$f"; FERN_GEMM_TILES=$f run; done
done
