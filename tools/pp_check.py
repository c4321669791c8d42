#!/usr/bin/env python
"""A/B of the ping-pong 256 x 256 GEMM tile (csrc/gemm_pp.h: cfg 7 of the bf16 family, 11 of the block-scaled fp8 family) against the
family's other configurations on the encoder shapes: bit-identity first, then microseconds per launch (hipEvents around `iters`
back-to-back launches) and TFLOP/s.
    python tools/pp_check.py [--family bf16|mx8|both] [--cfgs 0,1,2,7] [--shapes vit,text,big] [--iters 20] [--quant]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402

SHAPES = {
    "vit": [(12608, 3072, 768, 1), (12608, 768, 3072, 3), (12608, 2304, 768, 0), (12608, 768, 768, 3)],
    "text": [(4928, 2048, 512, 1), (4928, 512, 2048, 3), (4928, 1536, 512, 0), (4928, 512, 512, 3)],
    "big": [(4096, 4096, 4096, 0), (8192, 8192, 1024, 0)],
    "edge": [(300, 320, 128, 0), (1000, 260, 256, 1), (257, 511 // 32 * 32, 384, 3), (64, 512, 512, 0)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="both")
    ap.add_argument("--cfgs", default=None)
    ap.add_argument("--shapes", default="vit,big")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--quant", action="store_true", help="mx8: also the quantising epilogue on the bias / GELU shapes")
    args = ap.parse_args()
    eng = FernEngine("cuda:0")
    fams = ["bf16", "mx8"] if args.family == "both" else [args.family]
    for fam in fams:
        new = 7 if fam == "bf16" else 11
        cfgs = [int(c) for c in args.cfgs.split(",")] if args.cfgs else ([0, 1, 2, 6, 7] if fam == "bf16" else [0, 7, 9, 10, 11])
        for g in args.shapes.split(","):
            for (m, n, k, epi) in SHAPES[g]:
                if fam == "mx8" and k % 128:
                    continue
                gen = torch.Generator(device="cuda").manual_seed(m + n + k)
                a = torch.randn(m, k, device="cuda", generator=gen)
                w = torch.randn(n, k, device="cuda", generator=gen) * k ** -0.5
                b = torch.randn(n, device="cuda", generator=gen)
                r = torch.randn(m, n, device="cuda", generator=gen) if epi == 3 else None
                out_b = epi in (0, 1)
                runs = {}
                if fam == "bf16":
                    ab, wb = eng.to_bf16(a), eng.to_bf16(w)
                    runs["plain"] = lambda: eng.gemm_bf16(ab, wb, b, residual=r, epilogue=epi, out_bf16=out_b)
                else:
                    (a8, sa), (w8, sw) = eng.quantize_mx8(a), eng.quantize_mx8(w)
                    rb = eng.to_bf16(r) if r is not None else None
                    runs["plain"] = lambda: eng.gemm_mx8(a8, sa, w8, sw, b, residual=r, epilogue=epi, out_bf16=out_b)
                    if r is not None:
                        runs["bf16res"] = lambda: eng.gemm_mx8(a8, sa, w8, sw, b, residual=rb, epilogue=epi, out_bf16=True)
                    if args.quant and epi in (0, 1) and n % 128 == 0:
                        runs["quant"] = lambda: eng.gemm_mx8_quant(a8, sa, w8, sw, b, epilogue=epi)
                for rname, run in runs.items():
                    ref = None
                    line = f"{fam:5s} {rname:7s} {m:6d} {n:5d} {k:5d} epi {epi}:"
                    for c in cfgs:
                        eng.tuner_force_config(fam, c)
                        try:
                            out = run()
                        except RuntimeError as e:
                            line += f"  cfg{c}: refused"
                            continue
                        torch.cuda.synchronize()
                        outs = out if isinstance(out, tuple) else (out,)
                        flat = [o.view(torch.uint8).cpu() if o.dtype != torch.float32 else o.cpu() for o in outs]
                        if ref is None:
                            ref = flat
                            same = "ref"
                        else:
                            same = "same" if all(torch.equal(x, y) for x, y in zip(flat, ref)) else "DIFF"
                        for _ in range(3):
                            run()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(args.iters):
                            run()
                        e1.record()
                        torch.cuda.synchronize()
                        us = e0.elapsed_time(e1) / args.iters * 1e3
                        line += f"  cfg{c}{'*' if c == new else ''}: {us:7.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF {same}"
                    print(line, flush=True)
                    eng.tuner_force_config(fam, -1)


if __name__ == "__main__":
    main()
