#!/bin/bash
# bulk + 16x16-remainder pair plans of the fp32 family: plan test, text shapes tuned, c2 headline
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04pair
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mixed_geometry or test_gemm" > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 300 python tools/gemm_bench.py --shapes text > $O/gemm_bench.txt 2>&1; cat $O/gemm_bench.txt
timeout 600 python bench.py --no-cpu-baseline --headline-only --steps 20 --save-tiles $O/tiles.txt > $O/c2.json 2> $O/c2.err
grep "^f32 4928\|^f32 5824\|^f32 5040" $O/tiles.txt
python - <<PY
import json
j=json.loads([l for l in open("$O/c2.json") if l.startswith("{")][-1])
print("c2", round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms frac", round(j["roofline"]["frac"],4))
PY
