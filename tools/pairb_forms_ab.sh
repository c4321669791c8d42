#!/bin/bash
# Which form of each image + text GEMM pair is best with FOUR batches in flight (the pair tuner times isolated launches): c5 runs with the
# four `pairb` lines pinned to given forms (0 = two launches, 1 = 16-wave, 2 = 8-wave), order: out-proj, c_proj, QKV, c_fc.
#   bash tools/pairb_forms_ab.sh out.txt "1 1 1 1" "2 2 1 1" ...
OUT=${1:-gpurun_out/pairb_forms.txt}; shift
mkdir -p $(dirname $OUT); : > $OUT
run() {
  python bench.py --config c5 --headline-only --no-cpu-baseline --no-other-configs --steps 40 --full-record /tmp/fern_pf_full.json 2>/dev/null | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['value'],1), d['ms_per_step'])" >> $OUT
}
for rep in 1 2; do
  unset FERN_GEMM_TILES; run "tuner  "
  for forms in "$@"; do
    printf "pairb 12608 768 768 3 4 4928 512 512 3 0 %s\npairb 12608 768 3072 3 4 4928 512 2048 3 0 %s\npairb 12608 2304 768 0 5 4928 1536 512 0 1 %s\npairb 12608 3072 768 1 12 4928 2048 512 1 1 %s\n" $forms > /tmp/fern_pf_tiles.txt
    export FERN_GEMM_TILES=/tmp/fern_pf_tiles.txt; run "$forms"
  done
done
cat $OUT
