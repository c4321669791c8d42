#!/usr/bin/env python
"""Summarise a FERN_PROF_DUMP csv (kind,m,n,k,tag,ms,work) by shape: launches, total ms, TFLOP/s or GB/s."""
import collections
import sys

rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
for line in open(sys.argv[1]):
    kind, m, n, k, tag, ms, work = line.strip().split(",")
    key = (int(kind), int(m), int(n), int(k), int(tag))
    r = rows[key]
    r[0] += 1
    r[1] += float(ms)
    r[2] += float(work)
total = sum(r[1] for key, r in rows.items() if key[0] != 4)      # the stage interval contains its sweep: not added twice
print(f"{'kind':>4} {'M':>7} {'N':>6} {'K':>5} {'tag':>3} {'calls':>6} {'ms':>9} {'%':>6} {'rate':>10}")
names = {0: "gemm", 1: "attn", 2: "topk", 3: "swp", 4: "rank"}      # 4: the whole ranking stage (PROF_STAGE: its sweep is listed under 3 as well)
for key, (cnt, ms, work) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    rate = work / (ms * 1e-3) / 1e12 if ms > 0 else 0
    unit = "TB/s" if key[0] == 3 else "TF/s"
    print(f"{names.get(key[0], str(key[0])):>4} {key[1]:>7} {key[2]:>6} {key[3]:>5} {key[4]:>3} {cnt:>6} {ms:>9.3f} {100 * ms / total:>6.2f} {rate:>8.2f} {unit}")
