"""Helper of test_resnet_tower_is_bit_identical_on_every_lds_dma_tile: encodes two RN50x4 images with whatever tile FERN_GEMM_CFG
forces (3x3-window and 1x1 convolutions as GEMMs with N = 80 / 160 / 320 / ... columns) and saves the raw features.
Usage: python tests/_resnet_dump.py out.npy"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fashionern_aaai2024_amd import synth  # noqa: E402
from fashionern_aaai2024_amd.engine import FernEngine  # noqa: E402


def main():
    cfg = synth.CLIP_CONFIGS["RN50x4"]
    eng = FernEngine("cuda:0")
    eng.load_tensors(synth.clip_state_dict(cfg, seed=8))
    eng.finalize_clip(cfg)
    imgs = torch.from_numpy(synth.images(2, cfg)).cuda()
    np.save(sys.argv[1], eng.encode_image(imgs).cpu().numpy())


if __name__ == "__main__":
    main()
