#!/usr/bin/env python
"""fp32 (and, second block, bf16-operand) attention alone at the path's shapes (ViT 197 x 197, text 77 causal, fusion 91): time per launch and TFLOP/s.
`rocprofv3 --pmc ... -- python3 tools/attn_bench.py 5` feeds tools/pmc_kernel.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402

eng = FernEngine("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for name, b, heads, s, causal in (("vit", 64, 12, 197, False), ("text", 64, 8, 77, True), ("bert", 64, 8, 91, False)):
    w = heads * 64
    q, k, v = (torch.randn(b, s, w, device="cuda") for _ in range(3))
    for _ in range(3):
        eng.attention(q, k, v, heads, causal=causal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        eng.attention(q, k, v, heads, causal=causal)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 4.0 * b * heads * s * s * 64 * (0.5 if causal else 1.0)
    print(f"{name:5s} B={b} H={heads} S={s} causal={causal}: {us:7.1f} us/launch  {fl / us / 1e6:6.1f} TFLOP/s", flush=True)

for name, b, heads, s, causal in (("vit", 64, 12, 197, False), ("text", 64, 8, 77, True), ("bert", 64, 8, 91, False)):
    w = heads * 64
    q, k, v = (eng.to_bf16(torch.randn(b * s, w, device="cuda")).reshape(b, s, w) for _ in range(3))
    for _ in range(3):
        eng.attention_bf16(q, k, v, heads, causal=causal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        eng.attention_bf16(q, k, v, heads, causal=causal)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 4.0 * b * heads * s * s * 64 * (0.5 if causal else 1.0)
    by = 4.0 * b * s * w * 2
    print(f"bf16 {name:5s} B={b} H={heads} S={s} causal={causal}: {us:7.1f} us/launch  {fl / us / 1e6:6.1f} TFLOP/s  {by / us / 1e3:6.0f} GB/s", flush=True)
