#!/usr/bin/env python
"""Turn two `rocprofv3 --pmc` passes of `bench.py --pmc-mode` (one with FETCH_SIZE, one with WRITE_SIZE) into
profiles/pmc_traffic.json: HBM bytes per launch for the GEMM family and for the similarity sweep.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read (TCC_EA0_RDREQ x 64 B with 128-B requests tallied at 64 B) -> doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores (our epilogue stores 4 B per lane: uncalibrated, reported as is).

Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> [out.json]
"""
import csv
import json
import sys


def per_dispatch(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]), float(r["Counter_Value"])))
    rows.sort()
    marker = max(i for i, r in enumerate(rows) if "l2norm_kernel" in r[1] and r[2] == r[3])   # single-workgroup l2norm
    return rows[marker + 1:]


def summarise(rows, scale):
    gemm = [r for r in rows if "gemm_f32" in r[1]]
    sweep = [rows[i - 1] for i, r in enumerate(rows) if "topk_rows_kernel" in r[1] and i > 0 and "gemm_f32" in rows[i - 1][1]]
    sweep_ids = {r[0] for r in sweep}
    gemm = [r for r in gemm if r[0] not in sweep_ids]
    tot = lambda rs: sum(r[4] for r in rs) * 1024 * scale  # noqa: E731
    return {"gemm_launches": len(gemm), "gemm_bytes": tot(gemm), "sweep_launches": len(sweep), "sweep_bytes": tot(sweep)}


def main():
    fetch = summarise(per_dispatch(sys.argv[1], "FETCH_SIZE"), 2.0)
    write = summarise(per_dispatch(sys.argv[2], "WRITE_SIZE"), 1.0)
    steps = int(sys.argv[3])
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --pmc-mode`; FETCH_SIZE x2 (gfx950), KiB -> B",
        "steps": steps,
        "gemm": {"launches_per_step": fetch["gemm_launches"] / steps,
                 "fetch_bytes_per_launch": fetch["gemm_bytes"] / max(1, fetch["gemm_launches"]),
                 "write_bytes_per_launch": write["gemm_bytes"] / max(1, write["gemm_launches"])},
        "sweep": {"launches_per_step": fetch["sweep_launches"] / steps,
                  "fetch_bytes_per_launch": fetch["sweep_bytes"] / max(1, fetch["sweep_launches"]),
                  "write_bytes_per_launch": write["sweep_bytes"] / max(1, write["sweep_launches"])},
    }
    for k in ("gemm", "sweep"):
        out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
