#!/usr/bin/env python
"""Experiment: does running the (independent) text tower on a second HIP stream / context beside the ViT tower recover
the tile-quantisation tails of the GEMMs?  Prints ms per step serial vs overlapped."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd import synth  # noqa: E402
from fashionern_aaai2024_amd.clip_model import create_model  # noqa: E402
from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402
from fashionern_aaai2024_amd.model import ERN  # noqa: E402

cfg = synth.CLIP_CONFIGS["ViT-B-16"]
dev = torch.device("cuda:0")
csd = synth.clip_state_dict(cfg, 0)
clip = create_model(cfg, device=dev)
clip.load_state_dict(csd)
eng = clip.engine
model = ERN(clip, 512, dev, engine=eng).init_random(0)
eng2 = FernEngine(dev)
eng2.load_tensors({k: v for k, v in csd.items() if not k.startswith("visual.")})
import dataclasses
eng2.finalize_clip(dataclasses.replace(cfg, v_layers=0))
B = 64
images = torch.from_numpy(synth.images(B, cfg)).to(dev)
tokens = torch.from_numpy(synth.captions(B, cfg)).to(dev)
loc = torch.from_numpy(synth.local_feats(B, 512)).to(dev)
gallery = torch.from_numpy(synth.unit_rows(46000, 512)).to(dev)
side = torch.cuda.Stream()


def serial():
    rf = eng.encode_image(images)
    tg, ts = eng.encode_text(tokens)
    q = eng.dvr_fuse(rf, loc, tg, ts)
    return eng.sim_topk(q, gallery, 50)


def overlapped():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        tg, ts = eng2.encode_text(tokens)
    rf = eng.encode_image(images)
    main.wait_stream(side)
    tg.record_stream(main)
    ts.record_stream(main)
    q = eng.dvr_fuse(rf, loc, tg, ts)
    return eng.sim_topk(q, gallery, 50)


for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:10s} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms/step", flush=True)
a = serial()
b = overlapped()
torch.cuda.synchronize()
print("same top-k:", torch.equal(a[1], b[1]), "max score diff", (a[0] - b[0]).abs().max().item())

# ---- experiment 2: split the ViT batch over two streams / contexts (tails of one fill with the other's tiles) ----
eng3 = FernEngine(dev)
eng3.load_tensors(csd)
eng3.finalize_clip(cfg)


def vit_single():
    return eng.encode_image(images)


def vit_split():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        b_ = eng3.encode_image(images[32:])
    a_ = eng.encode_image(images[:32])
    main.wait_stream(side)
    b_.record_stream(main)
    return torch.cat((a_, b_))


for name, fn in (("vit single", vit_single), ("vit split2", vit_split), ("vit single", vit_single), ("vit split2", vit_split)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:10s} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms", flush=True)
print("identical:", torch.equal(vit_single(), vit_split()))
