#!/bin/bash
# round-end check on the GPU box: full GPU suite, PMC traffic of the c2 / c5 steps (stamped with the kernel-source hash), the driver's command.
#   bash tools/final_check.sh [outdir]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/${1:-gpurun_out/final}
mkdir -p $O
cd $R
if [ -z "${SKIP_SUITE:-}" ]; then
  python -m pytest tests -m gpu -x -q --durations=6 > $O/gpu_suite.log 2>&1
  tail -12 $O/gpu_suite.log
fi
export FERN_HEAD=$(cat $R/.fern_head 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c5; do
  # one set of tuner choices (tiles, pair forms) for both counter passes: they are separate runs, and a pair that is one launch in the
  # FETCH pass and two in the WRITE pass would mix two launch populations in the per-launch figure
  timeout 600 python3 $R/bench.py --no-cpu-baseline --headline-only --no-other-configs --lanes 1 --steps 4 --config $cfg --save-tiles /tmp/pmc_tiles_$cfg.txt --full-record /tmp/pmc_tiles_$cfg.json > /dev/null 2>&1
  export FERN_GEMM_TILES=/tmp/pmc_tiles_$cfg.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${cfg}_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${cfg}_$c -o p -- python3 $R/bench.py --pmc-mode --config $cfg --steps 2 --lanes 1 > /tmp/pmc_${cfg}_$c.log 2>&1
  done
  out=$O/pmc_traffic.json; [ $cfg = c5 ] && out=$O/pmc_traffic_c5.json
  unset FERN_GEMM_TILES
  python3 $R/tools/pmc_traffic.py $(find /tmp/pmc_${cfg}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_${cfg}_WRITE_SIZE -name "*counter_collection.csv" | head -1) 2 $out > $O/pmc_traffic_$cfg.log 2>&1
done
cd $R
cp $O/pmc_traffic.json $O/pmc_traffic_c5.json profiles/        # the driver's command below reads them (only on this box: copy them back in the dev container too)
( time python bench.py --gpus 1 --steps 20 --warmup 5 --full-record $O/bench_full.json > $O/bench.out 2> $O/bench.err ) 2> $O/bench.time
tail -c 1200 $O/bench.out; cat $O/bench.time
