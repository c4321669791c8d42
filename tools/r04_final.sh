#!/bin/bash
# end-of-round check on the committed tree: entry-point smoke, full GPU suite, default bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 600 $O/bench_c2.json
