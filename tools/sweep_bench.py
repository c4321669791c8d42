#!/usr/bin/env python
"""Ranking-stage roofline: fused sweep + top-K (sample pass, bound, filtered sweep, selection, gated exact pass) against
SURVEY 8d's algorithmic bytes N*D*s_g + B*D*4 + B*K*8, for an fp32 gallery (MFMA-bound at B = 64) and a bf16 gallery
(HBM-bound), per gallery size.  Whole-call time from HIP events on the stream; the sweep kernel alone from libfern's profiler."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402

eng = FernEngine("cuda:0")
D, K = 512, 50
sizes = [int(a) for a in sys.argv[1:]] or [46_000, 200_000, 1_000_000]
for B in (64, 128):
    for n in sizes:
        g = torch.nn.functional.normalize(torch.randn(n, D, device="cuda"), dim=-1)
        q = torch.nn.functional.normalize(torch.randn(B, D, device="cuda"), dim=-1)
        gb = eng.gallery_to_bf16(g)
        for name, sg, fn in (("fp32", 4, lambda: eng.sim_topk(q, g, K)), ("bf16", 2, lambda: eng.sim_topk_bf16(q, gb, K))):
            for _ in range(3):
                fn()
            iters = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            call_us = e0.elapsed_time(e1) / iters * 1e3
            eng.prof_enable(True)
            for _ in range(10):
                fn()
            st = eng.prof_collect()
            eng.prof_enable(False)
            nsweep = st["sweep_launches"] / 10
            sweep_us = st["sweep_ms"] / 10 * 1e3
            sel_us = st["topk_ms"] / 10 * 1e3
            alg = n * D * sg + B * D * 4 + B * K * 8
            print(f"B={B:4d} N={n:8d} {name}: call {call_us:8.1f} us = {alg / call_us / 1e3:6.0f} GB/s algorithmic ({alg / call_us / 1e3 / 8000:.3f} of 8 TB/s) | "
                  f"sweep kernel(s) {sweep_us:8.1f} us x{nsweep:.0f}, sample + bound + select + exact pass {sel_us:7.1f} us (instrumented)", flush=True)
        del g, gb
