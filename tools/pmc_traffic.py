#!/usr/bin/env python
"""Turn two `rocprofv3 --pmc` passes of `bench.py --pmc-mode` (one with FETCH_SIZE, one with WRITE_SIZE) into
profiles/pmc_traffic.json: HBM-side bytes per launch for the fp32 GEMM family and per call for the ranking stage
(sample GEMM + bound + filtered sweep + candidate select + the gated exact pass).

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read (TCC_EA0_RDREQ x 64 B with 128-B requests tallied at 64 B) -> doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores (the GEMM epilogues store 4 B per lane: uncalibrated, reported as is).

Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> [out.json]
"""
import csv
import datetime
import json
import os
import re
import sys


def per_dispatch(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]), float(r["Counter_Value"])))
    rows.sort()
    marker = max(i for i, r in enumerate(rows) if "l2norm_kernel" in r[1] and r[2] == r[3])   # single-workgroup l2norm
    return rows[marker + 1:]


def is_filter_gemm(name):      # gemm_f32_glds_kernel<BM, BN, WM, WN, BKT, MINW, CONV, SYNC, FILT = true, SPLIT>
    return re.search(r"gemm_f32_glds_kernel<(\d+,\s*){6}(true|false),\s*\d+,\s*true\b", name) is not None


RANK_KERNELS = ("sweep_bf16_kernel", "topk_sample_bound_kernel", "topk_candidates_kernel", "topk_rescore_kernel", "topk_dense_rescore_kernel",
                "topk_tiles_rescore_kernel", "rank_exact_kernel")


def summarise(rows, scale):
    """Per family: dispatches and counter bytes.  Ranking stage = every kernel of fern_sim_topk / _prefiltered / _bf16 (the bf16 sweeps,
    bound, select / rescore kernels, the gated exact pass; for the fp32-MFMA form also the sample GEMM in front of the bound kernel and
    the filtered GEMM sweep); sweep = the full-gallery pass of each call (bf16 sweep in its filter / store-all form, or the filtered GEMM)."""
    rank_ids, sweep_ids = set(), set()
    for i, r in enumerate(rows):
        name = r[1]
        if any(k in name for k in RANK_KERNELS) or is_filter_gemm(name):
            rank_ids.add(r[0])
        if "topk_sample_bound_kernel" in name and i > 0 and "gemm_f32" in rows[i - 1][1]:
            rank_ids.add(rows[i - 1][0])                          # fp32 form: the sample pass is the GEMM right before the bound kernel
        if is_filter_gemm(name):
            sweep_ids.add(r[0])
        if "sweep_bf16_kernel" in name:
            # the full pass of a call: the filter form, or the store-all form (the dense pre-filter: followed by topk_dense_rescore_kernel / topk_tiles_rescore_kernel)
            nxt = rows[i + 1][1] if i + 1 < len(rows) else ""
            if re.search(r"sweep_bf16_kernel<\d+,\s*true", name) or "topk_dense_rescore_kernel" in nxt or "topk_tiles_rescore_kernel" in nxt or "sweep_bf16_kernel" in nxt:
                sweep_ids.add(r[0])
    fam = lambda key: [r for r in rows if key in r[1] and r[0] not in rank_ids]  # noqa: E731
    # (gemm_mxbf_pair_kernel: a block-scaled image-tower GEMM carrying the text tower's bf16 GEMM of the same layer -- the mx8 family's launch)
    gemm, mx8, b16 = fam("gemm_f32"), fam("gemm_mx8_kernel") + fam("gemm_mxbf_pair_kernel"), fam("gemm_bf16_glds_kernel")
    rank = [r for r in rows if r[0] in rank_ids]
    sweep = [r for r in rows if r[0] in sweep_ids]
    # one select kernel per ranking call (the gated exact-pass launch no longer closes every call: the dense form on small galleries ranks
    # a query without room inside its select kernel)
    calls = sum(1 for r in rows if any(k in r[1] for k in ("topk_candidates_kernel", "topk_rescore_kernel", "topk_dense_rescore_kernel",
                                                            "topk_tiles_rescore_kernel")))
    tot = lambda rs: sum(r[4] for r in rs) * 1024 * scale  # noqa: E731
    return {"gemm_launches": len(gemm), "gemm_bytes": tot(gemm), "mx8_launches": len(mx8), "mx8_bytes": tot(mx8), "bf16_launches": len(b16),
            "bf16_bytes": tot(b16), "rank_calls": calls, "rank_bytes": tot(rank), "sweep_bytes": tot(sweep)}


def main():
    fetch = summarise(per_dispatch(sys.argv[1], "FETCH_SIZE"), 2.0)
    write = summarise(per_dispatch(sys.argv[2], "WRITE_SIZE"), 1.0)
    steps = int(sys.argv[3])
    calls = max(1, fetch["rank_calls"])
    # when / on which kernel sources the counters were collected: bench.py cannot collect PMC counters inside the driver's run, so it
    # reports `roofline.traffic` from this file ONLY when `csrc_sha16` (sha256 over csrc/*.hip, *.h -- bench.csrc_sha16) equals its own
    # tree's, and omits it otherwise.  FERN_HEAD: `git rev-parse --short HEAD` of the dev container (no .git on the GPU box).
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha16
    out = {
        "measured": {"date": datetime.date.today().isoformat(), "head": os.environ.get("FERN_HEAD", "unknown"), "csrc_sha16": csrc_sha16()},
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --pmc-mode`; FETCH_SIZE x2 (gfx950), KiB -> B",
        "steps": steps,
        "gemm": {"launches_per_step": fetch["gemm_launches"] / steps,
                 "fetch_bytes_per_launch": fetch["gemm_bytes"] / max(1, fetch["gemm_launches"]),
                 "write_bytes_per_launch": write["gemm_bytes"] / max(1, write["gemm_launches"])},
        "sweep": {"calls_per_step": fetch["rank_calls"] / steps,
                  "stage_fetch_bytes_per_call": fetch["rank_bytes"] / calls,
                  "stage_write_bytes_per_call": write["rank_bytes"] / max(1, write["rank_calls"]),
                  "sweep_kernel_fetch_bytes": fetch["sweep_bytes"] / calls,
                  "sweep_kernel_write_bytes": write["sweep_bytes"] / max(1, write["rank_calls"]),
                  "note": "stage = every kernel of one ranking call (fern_sim_topk_prefiltered on a prepared fp32 gallery: bf16 sweep + "
                          "select / rescore + gated exact pass; the dense form stores its [B, N] approximate scores, the list forms only "
                          "sample scores and candidates)"},
        "note": "FETCH_SIZE counts L2->fabric read requests (Infinity-Cache hits included), i.e. the sum over the 8 private XCD L2s: every XCD "
                "streams the weight panels of its tiles once per generation of resident tiles, so for the GEMM family this is L2-miss traffic "
                "(largely served by the 256 MiB Infinity Cache), not DRAM traffic.",
    }
    for fam_key, name in (("mx8", "gemm_mx8"), ("bf16", "gemm_bf16")):
        if fetch[fam_key + "_launches"]:
            n = fetch[fam_key + "_launches"]
            out[name] = {"launches_per_step": n / steps, "fetch_bytes_per_launch": fetch[fam_key + "_bytes"] / n,
                         "write_bytes_per_launch": write[fam_key + "_bytes"] / max(1, write[fam_key + "_launches"])}
            out[name]["hbm_bytes_per_launch"] = out[name]["fetch_bytes_per_launch"] + out[name]["write_bytes_per_launch"]
    out["gemm"]["hbm_bytes_per_launch"] = out["gemm"]["fetch_bytes_per_launch"] + out["gemm"]["write_bytes_per_launch"]
    out["sweep"]["hbm_bytes_per_launch"] = out["sweep"]["stage_fetch_bytes_per_call"] + out["sweep"]["stage_write_bytes_per_call"]
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
