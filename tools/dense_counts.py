#!/usr/bin/env python
"""Lab: how many rows the dense pre-filter kernel collects (FERN_DENSE_STOP=3 -> idx[:, 0] = collected) and rescoring survivors
(FERN_DENSE_STOP=5) per query at the C2 shape.  Run once per value of FERN_DENSE_STOP."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fashionern_aaai2024_amd.engine import FernEngine
eng = FernEngine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
n, d, b = 46000, 512, 64
gal = torch.nn.functional.normalize(torch.randn(n, d, generator=g, device="cuda"), dim=-1)
q = torch.nn.functional.normalize(torch.randn(b, d, generator=g, device="cuda"), dim=-1)
pg = eng.prepare_gallery(gal)
eng.set_rank_strategy("dense")
s, i = eng.sim_topk(q, pg, 50)
torch.cuda.synchronize()
print("stop", os.environ.get("FERN_DENSE_STOP"), "meta", pg.meta.tolist(), "idx[:,0] min/mean/max", i[:, 0].min().item(), i[:, 0].float().mean().item(), i[:, 0].max().item())
