#!/usr/bin/env python
"""One-stream timeline of the measured steps of `bench.py --pmc-mode` from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 bench.py --pmc-mode --config c5 --lanes 1 --steps 3
    python tools/step_timeline.py /tmp/tl 3 [--list]

The steps start behind the marker dispatch (a single-workgroup l2norm_kernel, bench.py --pmc-mode).  Prints, per kernel name
(template arguments kept, namespaces dropped): dispatches per step, us per step, share of the step's busy time; then the step's
span, the sum of the kernel durations and the idle time between kernels (launch boundaries).  --list: every dispatch of the
first measured step in order (start offset, duration, gap to the previous kernel's end, grid)."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        raise SystemExit(f"no kernel_trace.csv under {d}")
    rows = []
    for r in csv.DictReader(open(files[0])):
        def dim(prefix):
            if prefix in r:
                return int(r[prefix] or 0)
            v = 1
            for ax in "XYZ":
                v *= max(1, int(r.get(f"{prefix}_{ax}", 1) or 1))
            return v
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], dim("Grid_Size"), dim("Workgroup_Size")))
    rows.sort()
    marker = max(i for i, r in enumerate(rows) if "l2norm_kernel" in r[2] and r[3] == r[4])
    rows = rows[marker + 1:]
    if not rows:
        raise SystemExit("nothing behind the marker")
    short = lambda n: re.sub(r"\(.*$", "", n.replace("void ", "").replace("fern::", ""))[:96]  # noqa: E731
    agg = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    gaps = 0.0
    for i, (s, e, n, g, w) in enumerate(rows):
        a = agg[short(n)]
        a[0] += 1
        a[1] += (e - s) / 1e3
        busy += (e - s) / 1e3
        if i:
            gaps += max(0, s - rows[i - 1][1]) / 1e3
    span = (rows[-1][1] - rows[0][0]) / 1e3
    print(f"{len(rows)} dispatches in {steps} steps: span {span / steps:.1f} us/step, kernels {busy / steps:.1f} us/step, "
          f"idle between kernels {gaps / steps:.1f} us/step ({len(rows) / steps:.0f} dispatches/step)")
    print(f"{'kernel':96s} {'n/step':>7s} {'us/step':>9s} {'avg us':>8s} {'%':>6s}")
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:96s} {c / steps:7.1f} {us / steps:9.1f} {us / c:8.1f} {100 * us / busy:6.2f}")
    if "--list" in sys.argv:
        t0 = rows[0][0]
        per = len(rows) // steps
        print("\n# first measured step, in order: start us, duration us, gap before us, grid/wg, kernel")
        for i, (s, e, n, g, w) in enumerate(rows[:per]):
            gap = (s - rows[i - 1][1]) / 1e3 if i else 0.0
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {g // max(w, 1):6d}x{w:<4d} {short(n)}")


if __name__ == "__main__":
    main()
