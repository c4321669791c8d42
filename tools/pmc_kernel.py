#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name (first dispatch of each name x grid): clock, MFMA pipe
utilisation, wait split, LDS activity / bank conflicts.  usage: pmc_kernel.py <counter_collection.csv> [name filter]"""
import collections
import csv
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if flt not in r["Kernel_Name"]:
        continue
    d = rows[int(r["Dispatch_Id"])]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["name"] = r["Kernel_Name"].split("(")[0].replace("void fern::", "")
    d["grid"] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(list)
for did in sorted(rows):
    d = rows[did]
    agg[(d["name"], d["grid"])].append(d)
for (name, grid), ds in agg.items():
    d = ds[len(ds) // 2]
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    clk = cyc / d["dur"] / 1e3 if d["dur"] and cyc else 0
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    out = f"{name[:60]:60s} blocks={grid:6d} n={len(ds):3d} {d['dur']:8.1f}us"
    if cyc:
        out += f" clk={clk:5.2f}GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            out += f" mfma_util={d['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.2f}"
        if "SQ_LDS_IDX_ACTIVE" in d:
            out += f" lds_active={d['SQ_LDS_IDX_ACTIVE'] / (cyc * 256):5.2f} lds_conflict={d.get('SQ_LDS_BANK_CONFLICT', 0) / (cyc * 256):5.2f}"
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU",
              "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES", "SQ_WAVES"):
        if k in d:
            out += f" {k[3:].lower()}={d[k] / wc:6.3f}" if k not in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES") else f" {k[3:].lower()}={d[k]:.0f}"
    print(out)
