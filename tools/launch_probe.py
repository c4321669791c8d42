#!/usr/bin/env python
"""How much of a step is host-side launch time?  Enqueue time of N steps (no sync) vs their GPU completion time,
eager launches vs hipGraph replay, per encoder precision.  Usage: python tools/launch_probe.py [--steps 20]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd import synth  # noqa: E402
from fashionern_aaai2024_amd.clip_model import create_model  # noqa: E402
from fashionern_aaai2024_amd.model import ERN  # noqa: E402
from fashionern_aaai2024_amd.pipeline import ComposedQueryPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lanes", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = synth.CLIP_CONFIGS["ViT-B-16"]
    clip = create_model(cfg, device=dev, seed=0)
    model = ERN(clip, 512, dev, engine=clip.engine).init_random(0)
    eng = model.engine
    images = torch.from_numpy(synth.images(64, cfg, 42)).to(dev)
    tokens = torch.from_numpy(synth.captions(64, cfg, 42)).to(dev)
    loc = torch.from_numpy(synth.local_feats(64, 512, 42)).to(dev)
    gallery = torch.nn.functional.normalize(torch.randn(46000, 512, device=dev), dim=-1)
    pipe = ComposedQueryPipeline(eng, lanes=args.lanes)
    for prec in ("fp32", "bf16"):
        pipe.set_precision(prec)
        for _ in range(6):
            pipe.submit(images, tokens, loc, gallery, 50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.submit(images, tokens, loc, gallery, 50)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{prec}: enqueue {1e3 * (t1 - t0) / args.steps:.2f} ms/step, complete {1e3 * (t2 - t0) / args.steps:.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
