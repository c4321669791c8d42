#!/bin/bash
# A/B of the mixed mode's image + text GEMM pair (gemm_bf16.hip: launch_gemm_mxbf_pair) on the c5 workload: three alternating runs with the pair
# launcher on and off (FERN_GEMM_PAIR=0: two launches per pair), same box.    bash tools/pairb_ab.sh out.txt
OUT=${1:-gpurun_out/pairb_ab.txt}
: > "$OUT"
for rep in 1 2 3; do
  for pair in 1 0; do
    FERN_GEMM_PAIR=$pair python bench.py --config c5 --headline-only --no-cpu-baseline --no-other-configs --steps 40 --full-record /tmp/fern_pairb_full.json \
      --save-tiles /tmp/fern_pairb_tiles_$pair.txt 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('pair=$pair rep=$rep value', round(d['value'], 1), 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> "$OUT"
  done
done
echo "--- pair choices (0 = two launches, 1 = 16-wave pair kernel, 2 = 8-wave)" >> "$OUT"
grep '^pairb' /tmp/fern_pairb_tiles_1.txt >> "$OUT"
cat "$OUT"
