#!/bin/bash
# c5 throughput against the number of batches in flight (profiles/r06_c5_lanes.txt, r06_lanes_hwq.txt): bash tools/c5_probe.sh
O=gpurun_out/r06e; mkdir -p $O
for L in 2 3 4 6; do
  timeout 300 python bench.py --config c5 --headline-only --no-cpu-baseline --no-other-configs --steps 60 --lanes $L 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lanes', $L, round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['latency_ms_per_batch'])"
done | tee $O/c5_lanes.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_c5
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_c5 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-mode --config c5 --lanes 1 --steps 3 > /tmp/tl_c5.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/tl_c5 3 --list > $GRAFT_REPO_ROOT/$O/timeline_c5_mx8img.txt 2>&1
head -45 $GRAFT_REPO_ROOT/$O/timeline_c5_mx8img.txt | cut -c1-160
