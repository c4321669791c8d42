#!/usr/bin/env python
"""Lab: where the drop-in harness's host time goes on the GPU box -- DataLoader iteration alone (0 / 4 workers, with the GPU context
alive in the parent as in a real run), tokenising, the engine submits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
from torch.utils.data import DataLoader
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
sys.argv = ["bench.py"]
bench = importlib.util.module_from_spec(spec); sys.modules["bench"] = bench; spec.loader.exec_module(bench)
from fashionern_aaai2024_amd import synth
from fashionern_aaai2024_amd.engine import FernEngine
from fashionern_aaai2024_amd.utils import collate_fn, host_threads
eng = FernEngine("cuda:0")
x = torch.randn(1 << 28, device="cuda")      # 1 GiB resident like a gallery
cfg = synth.CLIP_CONFIGS["ViT-B-16"]
D = 512
host_local = torch.from_numpy(synth.local_feats(256, D, 31, "bench-ql"))
rel = bench._BenchRelativeDataset(2048, 46000, host_local)
pool_im = torch.from_numpy(synth.images(32, cfg, 9)); pool_lc = torch.from_numpy(synth.local_feats(32, D, 9, "bench-il"))
ds = bench._BenchIndexDataset(2048, pool_im, pool_lc)
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count(), flush=True)
for cap in (None, 16):
    for name, dset, bs in (("relative", rel, 64), ("index", ds, 32)):
        for nw in (0, 4):
            for pin in (False, True):
                ctx = host_threads(cap) if cap else host_threads(10 ** 6)
                with ctx:
                    t0 = time.perf_counter()
                    loader = DataLoader(dset, batch_size=bs, num_workers=nw, pin_memory=pin, collate_fn=collate_fn)
                    it = iter(loader)
                    first = next(it)
                    t1 = time.perf_counter()
                    n = 1
                    for _ in it:
                        n += 1
                    t2 = time.perf_counter()
                print(f"cap={cap} {name} workers={nw} pin={pin}: first batch {1e3 * (t1 - t0):.1f} ms, then {1e3 * (t2 - t1) / max(1, n - 1):.2f} ms/batch ({n} batches)", flush=True)
