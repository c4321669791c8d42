#!/usr/bin/env python
"""Experiment: N independent lanes (own HIP stream + own native contexts), consecutive 64-query steps dealt round-robin
to the lanes, so one lane's low-occupancy kernels (M=64 fusion GEMMs, tile tails, top-K) overlap the other's encoders."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd import synth  # noqa: E402
from fashionern_aaai2024_amd.clip_model import create_model  # noqa: E402
from fashionern_aaai2024_amd.model import ERN  # noqa: E402

cfg = synth.CLIP_CONFIGS["ViT-B-16"]
dev = torch.device("cuda:0")
csd = synth.clip_state_dict(cfg, 0)
fsd = synth.fusion_state_dict(512, 0)
B = 64
images = torch.from_numpy(synth.images(B, cfg)).to(dev)
tokens = torch.from_numpy(synth.captions(B, cfg)).to(dev)
loc = torch.from_numpy(synth.local_feats(B, 512)).to(dev)
gallery = torch.from_numpy(synth.unit_rows(46000, 512)).to(dev)


class Lane:
    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.clip = create_model(cfg, device=dev)
        self.clip.load_state_dict(csd)
        self.eng = self.clip.engine
        self.model = ERN(self.clip, 512, dev, engine=self.eng).load_state_dict(fsd)

    def step(self):
        with torch.cuda.stream(self.stream):
            rf = self.eng.encode_image(images)
            tg, ts = self.eng.encode_text(tokens)
            q = self.eng.dvr_fuse(rf, loc, tg, ts)
            return self.eng.sim_topk(q, gallery, 50)


for nl in (1, 2, 3):
    lanes = [Lane() for _ in range(nl)]
    for ln in lanes:
        for _ in range(2):
            ln.step()
    torch.cuda.synchronize()
    steps = 12
    t0 = time.perf_counter()
    for i in range(steps):
        out = lanes[i % nl].step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"lanes={nl}: {dt * 1e3:8.3f} ms/step  {B / dt:8.1f} q/s", flush=True)
    del lanes
