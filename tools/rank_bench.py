#!/usr/bin/env python
"""Ranking-stage micro-benchmark: fern_sim_topk (fp32-MFMA sweep) vs fern_sim_topk_prefiltered (certified bf16 pre-filter + exact fp32
rescoring) at the BASELINE shapes, timed by libfern's own instrumentation (one stream-marker interval around the WHOLE stage, the
sweep kernel by its dispatch timestamps) and by wall clock over back-to-back calls.  Results must be identical (asserted).

    python tools/rank_bench.py [--reps 20]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402

SHAPES = [("c2", 64, 46_000, 512, 50), ("c3", 64, 200_000, 640, 50), ("c4", 128, 21_552, 512, 51), ("c4x8", 1024, 21_552, 512, 51),
          ("1M", 64, 1_000_000, 512, 50), ("c3shard", 64, 25_000, 640, 50)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=str, default=None)
    ap.add_argument("--trace", type=str, default=None,
                    help="SHAPE:fp32_sweep|prefiltered -- for `rocprofv3 --kernel-trace`: warm up, emit the marker dispatch tools/step_timeline.py "
                         "keys on, run --reps calls of that one variant, exit")
    args = ap.parse_args()
    eng = FernEngine("cuda:0")
    dev = eng.device
    out = {}
    for name, b, n, d, k in SHAPES:
        if args.only and name not in args.only.split(","):
            continue
        g = torch.Generator(device=dev).manual_seed(n + d)
        gal = torch.nn.functional.normalize(torch.randn(n, d, generator=g, device=dev), dim=-1)
        q = torch.nn.functional.normalize(torch.randn(b, d, generator=g, device=dev), dim=-1)
        t0 = time.perf_counter()
        pg = eng.prepare_gallery(gal)
        torch.cuda.synchronize()
        prep_ms = (time.perf_counter() - t0) * 1e3
        if args.trace:
            tname, variant = args.trace.split(":")
            if tname != name:
                continue
            gg = pg if variant.startswith("prefiltered") else gal
            eng.set_rank_strategy({"prefiltered_lists": "lists", "prefiltered_dense": "dense"}.get(variant, "auto"))
            for _ in range(3):
                eng.sim_topk(q, gg, k)
            torch.cuda.synchronize()
            eng.l2_normalize(torch.zeros(3, 64, device=dev))
            for _ in range(args.reps):
                eng.sim_topk(q, gg, k)
            torch.cuda.synchronize()
            return
        rec = {"B": b, "N": n, "D": d, "K": k, "prepare_ms_first_call": prep_ms, "algorithmic_bytes": n * d * 4 + b * d * 4 + b * k * 8}
        res = {}
        for label, gg in (("fp32_sweep", gal), ("prefiltered_lists", pg), ("prefiltered_dense", pg), ("prefiltered", pg)):
            eng.set_rank_strategy({"prefiltered_lists": "lists", "prefiltered_dense": "dense"}.get(label, "auto"))
            for _ in range(3):
                res[label] = eng.sim_topk(q, gg, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                eng.sim_topk(q, gg, k)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / args.reps * 1e6
            eng.prof_enable(True)
            for _ in range(args.reps):
                eng.sim_topk(q, gg, k)
            st = eng.prof_collect()
            eng.prof_enable(False)
            calls = max(1, (b + 1023) // 1024) * args.reps
            stage = (st["sweep_ms"] + st["topk_ms"]) / calls * 1e3
            rec[label] = {"stage_us": stage, "sweep_kernel_us": st["sweep_ms"] / max(1, st["sweep_launches"]) * 1e3,
                          "sweep_launches_per_call": st["sweep_launches"] / args.reps, "wall_us_per_call": wall,
                          "stage_GBs_of_algorithmic_bytes": rec["algorithmic_bytes"] / (stage * 1e-6) / 1e9 * max(1, (b + 1023) // 1024),
                          "frac_of_8TBs": rec["algorithmic_bytes"] / (stage * 1e-6) / 1e9 / 8000.0 * max(1, (b + 1023) // 1024)}
        same = all(torch.equal(res["fp32_sweep"][0], res[v][0]) and torch.equal(res["fp32_sweep"][1], res[v][1])
                   for v in ("prefiltered_lists", "prefiltered_dense", "prefiltered"))
        rec["identical"] = bool(same)
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
        del gal, pg
        torch.cuda.empty_cache()
    bad = [k for k, v in out.items() if not v["identical"]]
    if bad:
        raise SystemExit(f"prefiltered ranking differs from the fp32 sweep at {bad}")


if __name__ == "__main__":
    main()
