#!/usr/bin/env python
"""Similarity-sweep roofline: fp32 gallery (MFMA-bound at B=64) vs bf16 gallery (HBM-bound), per gallery size.
Reports the kernel-only time of the sweep (HIP events via libfern's profiler) and algorithmic GB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402

eng = FernEngine("cuda:0")
B, D, K = 64, 512, 50
for n in (46_000, 200_000, 1_000_000):
    g = torch.nn.functional.normalize(torch.randn(n, D, device="cuda"), dim=-1)
    q = torch.nn.functional.normalize(torch.randn(B, D, device="cuda"), dim=-1)
    gb = eng.gallery_to_bf16(g)
    for name, fn in (("fp32", lambda: eng.sim_topk(q, g, K)), ("bf16", lambda: eng.sim_topk_bf16(q, gb, K))):
        for _ in range(3):
            fn()
        eng.prof_enable(True)
        for _ in range(10):
            fn()
        st = eng.prof_collect()
        eng.prof_enable(False)
        us = st["sweep_ms"] / st["sweep_launches"] * 1e3
        gbs = st["sweep_bytes"] / st["sweep_launches"] / (us * 1e-6) / 1e9
        gal_gbs = n * D * (4 if name == "fp32" else 2) / (us * 1e-6) / 1e9
        print(f"N={n:8d} {name}: sweep {us:9.1f} us  algorithmic {gbs:7.0f} GB/s (gallery alone {gal_gbs:7.0f} GB/s)  "
              f"top-K {st['topk_ms'] / st['topk_launches'] * 1e3:7.1f} us", flush=True)
