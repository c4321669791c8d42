#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc run of tools/gemm_bench.py: per GEMM dispatch clock, MFMA utilisation, wait split."""
import collections
import csv
import sys

rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_" not in r["Kernel_Name"]:
        continue
    d = rows[int(r["Dispatch_Id"])]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["name"] = r["Kernel_Name"].split("(")[0].replace("void fern::", "")
    d["grid"] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
seen = set()
for did in sorted(rows):
    d = rows[did]
    key = (d["name"], d["grid"])
    if key in seen:
        continue
    seen.add(key)
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    clk = cyc / d["dur"] / 1e3 if d["dur"] else 0
    util = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else 0
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(f"{d['name']:48s} blocks={d['grid']:6d} {d['dur']:8.1f}us clk={clk:5.2f}GHz mfma_util={util:5.2f} "
          f"wait_any={d.get('SQ_WAIT_ANY', 0) / wc:5.2f} wait_inst={d.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f} active={d.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.2f}"
          + (f" lds_active={d['SQ_LDS_IDX_ACTIVE'] / (cyc * 256):5.2f} lds_conflict={d.get('SQ_LDS_BANK_CONFLICT', 0) / (cyc * 256):5.2f}"
             if "SQ_LDS_IDX_ACTIVE" in d and cyc else ""))
