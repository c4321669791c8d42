#!/usr/bin/env python
"""Micro-benchmark of libfern's fp32 MFMA GEMM on the shapes of the hot path (A/B tool for kernel work).
Usage: python tools/gemm_bench.py [--iters 20] [--shapes vit|text|fusion|all]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402

SHAPES = {
    "vit": [(12608, 3072, 768, 1), (12608, 768, 3072, 3), (12608, 2304, 768, 0), (12608, 768, 768, 3)],
    "text": [(4928, 2048, 512, 1), (4928, 512, 2048, 3), (4928, 1536, 512, 0), (4928, 512, 512, 3)],
    "fusion": [(5824, 3072, 512, 1), (5824, 512, 3072, 3), (5824, 1536, 512, 0), (64, 4096, 4096, 2), (64, 2048, 512, 2),
               (64, 46000, 512, 0), (832, 512, 512, 0)],
    "big": [(4096, 4096, 4096, 0), (8192, 8192, 1024, 0)],
    "stride": [(64, 4096, 4096, 2), (64, 4096, 4032, 2), (64, 4096, 4160, 2), (64, 4096, 3072, 2), (64, 4096, 2048, 2)],
    "gelu": [(12608, 3072, 768, 0), (12608, 3072, 768, 1), (12608, 3072, 768, 2)],
    "ab": [(12608, 2304, 768, 0), (12608, 3072, 768, 1), (8192, 8192, 1024, 0)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--bf16", action="store_true", help="bf16-operand kernel (FERN_GEMM_BF16_CFG picks the tile)")
    ap.add_argument("--fp8", action="store_true", help="per-row-scaled fp8 kernel (FERN_GEMM_FP8_CFG)")
    ap.add_argument("--mx8", action="store_true", help="block-scaled fp8 kernel (FERN_GEMM_MX8_CFG)")
    ap.add_argument("--mx8q", action="store_true", help="with --mx8: quantising epilogue (fp8 + block scales out) on the BIAS / GELU shapes")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f32x3"], help="fp32 family arithmetic (FERN_GEMM_SPLIT_CFG picks the f32x3 tile)")
    args = ap.parse_args()
    eng = FernEngine("cuda:0")
    eng.set_precision(args.precision)
    groups = SHAPES if args.shapes == "all" else {args.shapes: SHAPES[args.shapes]}
    tot_ms = tot_fl = 0.0
    for gname, shapes in groups.items():
        for (m, n, k, epi) in shapes:
            a = torch.randn(m, k, device="cuda")
            w = torch.randn(n, k, device="cuda") * k ** -0.5
            b = torch.randn(n, device="cuda")
            r = torch.randn(m, n, device="cuda") if epi == 3 else None
            if args.fp8 or args.mx8:
                if epi == 2 or k % 128:
                    continue
                quant, gemm = (eng.quantize_mx8, eng.gemm_mx8) if args.mx8 else (eng.quantize_rows_fp8, eng.gemm_fp8)
                (a8, sa), (w8, sw) = quant(a), quant(w)
                out_b = epi in (0, 1)
                run = lambda: gemm(a8, sa, w8, sw, b, residual=r, epilogue=epi, out_bf16=out_b)  # noqa: E731
                if args.mx8 and args.mx8q and epi in (0, 1):
                    run = lambda: eng.gemm_mx8_quant(a8, sa, w8, sw, b, epilogue=epi)  # noqa: E731
            elif args.bf16:
                ab, wb = eng.to_bf16(a), eng.to_bf16(w)
                out_b = epi in (0, 1)
                run = lambda: eng.gemm_bf16(ab, wb, b, residual=r, epilogue=epi, out_bf16=out_b)  # noqa: E731
            else:
                run = lambda: eng.gemm(a, w, b, residual=r, epilogue=epi)  # noqa: E731
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            fl = 2.0 * m * n * k
            tot_ms += ms
            tot_fl += fl
            print(f"{gname:7s} M={m:6d} N={n:6d} K={k:5d} epi={epi}  {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)
    print(f"total {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
