#!/usr/bin/env python
"""Reference point for the hand-written GEMMs: what the vendor libraries (rocBLAS / hipBLASLt, through torch.matmul + bias) reach
on the same shapes on this GPU.  Plain C = A W^T (+ bias): no GELU / residual fusion, so the library numbers are an upper
bound for it on the fused shapes.  Usage: python tools/library_gemm_ref.py [--iters 30]"""
import argparse

import torch

SHAPES = [(12608, 3072, 768), (12608, 768, 3072), (12608, 2304, 768), (12608, 768, 768), (4928, 2048, 512), (4928, 512, 2048),
          (4096, 4096, 4096), (8192, 8192, 1024), (64, 4096, 4096)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    for dtype in (torch.float32, torch.bfloat16):
        for (m, n, k) in SHAPES:
            a = torch.randn(m, k, device="cuda", dtype=dtype)
            w = torch.randn(n, k, device="cuda", dtype=dtype) * k ** -0.5
            b = torch.randn(n, device="cuda", dtype=dtype)
            for _ in range(3):
                torch.nn.functional.linear(a, w, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                torch.nn.functional.linear(a, w, b)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            print(f"{str(dtype).split('.')[-1]:9s} M={m:6d} N={n:6d} K={k:5d}  {ms * 1e3:9.1f} us  {2.0 * m * n * k / ms / 1e9:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
