#!/usr/bin/env python
"""Print the top rows of a rocprofv3 `--kernel-trace --stats` kernel_stats.csv found under a directory.
Usage: python tools/kstats.py <dir> [rows] [copy_to]"""
import csv
import glob
import os
import shutil
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
if not files:
    raise SystemExit(f"no kernel_stats.csv under {d}: {os.listdir(d) if os.path.isdir(d) else 'missing dir'}")
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("fern::", "")[:72]
    print("%-72s calls=%6d total_ms=%9.2f avg_us=%8.1f pct=%5.1f" % (
        name, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
if len(sys.argv) > 3:
    shutil.copy(files[0], sys.argv[3])
