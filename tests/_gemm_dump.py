"""Helper of test_small_m_16x16_kernel_is_bit_identical_to_the_32x32_kernels: computes a fixed set of small-M GEMMs and fused
fusion-stage outputs with whatever FERN_GEMM_CFG forces and saves the raw results.  Usage: python tests/_gemm_dump.py out.npz"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fashionern_aaai2024_amd import synth  # noqa: E402
from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402


def main():
    eng = FernEngine("cuda:0")
    out = {}
    g = torch.Generator().manual_seed(7)
    for (m, n, k) in [(64, 512, 512), (64, 4096, 4096), (1, 256, 64), (17, 96, 128), (128, 768, 3072), (33, 1000, 96)]:
        a = torch.randn(m, k, generator=g)
        w = torch.randn(n, k, generator=g) * k ** -0.5
        b = torch.randn(n, generator=g)
        r = torch.randn(m, n, generator=g)
        for epi in (0, 1, 2, 3):
            out[f"gemm_{m}_{n}_{k}_{epi}"] = eng.gemm(a, w, b, residual=r if epi == 3 else None, epilogue=epi).cpu().numpy()
    d = 128
    eng.load_tensors(synth.fusion_state_dict(d, seed=4))
    eng.finalize_fusion(d)
    for b_ in (1, 7, 64):
        rg = torch.from_numpy(synth.global_feats(b_, d, tag="dg"))
        rl = torch.from_numpy(synth.local_feats(b_, d, tag="dl"))
        tg = torch.from_numpy(synth.global_feats(b_, d, tag="dt"))
        ts = torch.from_numpy(synth._normal(3, "dts", (b_, 77, d)))
        out[f"dvr_{b_}"] = eng.dvr_fuse(rg, rl, tg, ts).cpu().numpy()
        out[f"index_{b_}"] = eng.index_fuse(rg, rl, normalize_input=True).cpu().numpy()
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    main()
