#!/bin/bash
# phase attribution of the dense select kernel (profiles/r05_dense_phases.txt): the kernel ended after phase k by the lab switch FERN_DENSE_STOP
cd /tmp && export TMPDIR=/tmp
for st in 1 2 3 36 5 0; do
  FERN_DENSE_STOP=$st rocprofv3 --kernel-trace --output-format csv -d /tmp/tls_$st -o p -- python3 $GRAFT_REPO_ROOT/tools/rank_bench.py --reps 10 --trace c2:prefiltered_dense > /dev/null 2>&1
  echo "stop=$st" >> $GRAFT_REPO_ROOT/gpurun_out/r05_dense_phases.txt
  python3 $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/tls_$st 10 | grep "dense_rescore" >> $GRAFT_REPO_ROOT/gpurun_out/r05_dense_phases.txt
done
