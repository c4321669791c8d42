#!/bin/bash
# fp32 headline (c2), four batches in flight: the four image + text GEMM pairs (gemm.hip: launch_gemm_pair) pinned to one launch (1) or two (0),
# order out-proj, c_proj, QKV, c_fc -- the pair tuner times isolated launches.   bash tools/pair_forms_ab.sh out.txt "1 1 1 1" "1 0 1 1" ...
OUT=${1:-gpurun_out/pair_forms.txt}; shift
mkdir -p $(dirname $OUT); : > $OUT
run() {
  python bench.py --headline-only --no-cpu-baseline --no-other-configs --steps 40 --full-record /tmp/fern_pf_full.json 2>/dev/null | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['value'],1), d['ms_per_step'], d['roofline']['frac'])" >> $OUT
}
for rep in 1 2; do
  unset FERN_GEMM_TILES; run "tuner  "
  for forms in "$@"; do
    printf "pair 12608 768 768 3 0 4928 512 512 3 0 %s\npair 12608 768 3072 3 0 4928 512 2048 3 0 %s\npair 12608 2304 768 0 0 4928 1536 512 0 0 %s\npair 12608 3072 768 1 0 4928 2048 512 1 0 %s\n" $forms > /tmp/fern_pf_tiles.txt
    export FERN_GEMM_TILES=/tmp/fern_pf_tiles.txt; run "$forms"
  done
done
cat $OUT
