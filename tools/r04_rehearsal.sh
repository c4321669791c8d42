#!/bin/bash
# world = 4 rehearsal on the one GPU of a dev box (gloo, ranks share the GPU) of this round's bench.py: pre-flight, sharded gallery build,
# gathers, tuner-plan broadcast, sharded merge (c5).  Not a performance measurement.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04reh
mkdir -p $O
export FERN_BENCH_SHARE_GPU=1 FERN_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for c in c2 c5; do
  timeout 900 python bench.py --gpus 4 --config $c --steps 6 --warmup 3 --headline-only --no-cpu-baseline \
     > $O/r04_bench_${c}_4ranks_one_gpu_gloo.json 2> $O/r04_bench_${c}_4ranks.err
  echo "$c rc=$?"
  python - <<PY
import json
j=json.loads([l for l in open("$O/r04_bench_${c}_4ranks_one_gpu_gloo.json") if l.startswith("{")][-1])
print(j["n_gpus"], round(j["value"]), j["rccl_world"], (j.get("preflight") or {}).get("all_gather_64MiB_per_rank", {}).get("payload_ok"), (j.get("sharded_merge") or {}).get("identical_to_replicated"))
PY
done
