#!/bin/bash
# A/B of the block-scaled GEMM tile choice under the c5 pipeline's concurrency (3 batches in flight): the per-shape tuner times isolated
# launches; this compares its choices with one forced configuration on the SAME box, alternating.   bash tools/c5_cfg_ab.sh [cfgs...]
run() { timeout 300 python bench.py --config c5 --headline-only --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(\"{\")][-1]); print(round(d[\"value\"]), round(d[\"ms_per_step\"],3), round(d[\"roofline\"][\"frac\"],4))"; }
for rep in 1 2 3; do
  echo "tuned"; run
  for c in ${@:-7}; do echo "mx8 cfg $c"; FERN_GEMM_MX8_CFG=$c run; done
done
