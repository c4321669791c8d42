#!/usr/bin/env python
"""Images/s (and captions/s) of a CLIP tower configuration on one GPU, with a per-shape GEMM breakdown.
Usage: python tools/tower_bench.py RN50x4 [--batch 64]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd import synth  # noqa: E402
from fashionern_aaai2024_amd.clip_model import create_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
cfg = synth.CLIP_CONFIGS[a.name]
clip = create_model(cfg, device="cuda:0", seed=0)
eng = clip.engine
imgs = torch.from_numpy(synth.images(a.batch, cfg)).cuda()
toks = torch.from_numpy(synth.captions(a.batch, cfg)).cuda()
for fn, label, arg in ((eng.encode_image, "images", imgs), (eng.encode_text, "captions", toks)):
    for _ in range(2):
        fn(arg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn(arg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    eng.prof_enable(True)
    fn(arg)
    st = eng.prof_collect()
    eng.prof_enable(False)
    print(f"{a.name} {label}: {a.batch / dt:9.1f} /s  ({dt * 1e3:.2f} ms per batch of {a.batch}); GEMM {st['gemm_flops'] / 1e9:.0f} GFLOP "
          f"at {st['gemm_flops'] / max(st['gemm_ms'], 1e-9) / 1e9:.1f} TFLOP/s ({st['gemm_ms']:.2f} ms), attention {st['attn_ms']:.2f} ms")
