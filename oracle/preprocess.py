"""Oracle for the image side: the reference's preprocessing run with the real PIL (Pillow is the third-party library the
arithmetic lives in; torchvision is absent, its Resize / CenterCrop / ToTensor / Normalize semantics on PIL images are
restated from the call sites).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/dataloader/dataset.py:31-87 and /root/reference/utils/extract_fashioniq_patch.py:18-44,142-156.
"""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image

MEAN = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(3, 1, 1)
STD = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(3, 1, 1)


def target_pad(img: Image.Image, target_ratio: float) -> Image.Image:                 # dataset.py:46-54
    w, h = img.size
    if max(w, h) / min(w, h) < target_ratio:
        return img
    scaled = max(w, h) / target_ratio
    hp, vp = max(int((scaled - w) / 2), 0), max(int((scaled - h) / 2), 0)
    out = Image.new(img.mode, (w + 2 * hp, h + 2 * vp), 0)                              # F.pad(image, padding, 0, 'constant')
    if img.mode == "P":                                                                 # torchvision keeps the palette of P-mode images
        out.putpalette(img.getpalette())
    out.paste(img, (hp, vp))
    return out


def targetpad_transform(img: Image.Image, target_ratio: float = 1.25, dim: int = 288) -> torch.Tensor:   # dataset.py:73-87
    img = target_pad(img, target_ratio)
    w, h = img.size
    size = (dim, int(dim * h / w)) if w <= h else (int(dim * w / h), dim)               # torchvision Resize(int)
    img = img.resize(size, Image.BICUBIC)
    w, h = img.size
    top, left = int(round((h - dim) / 2.0)), int(round((w - dim) / 2.0))               # CenterCrop
    img = img.crop((left, top, left + dim, top + dim)).convert("RGB")
    x = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).to(torch.float32).div(255)   # ToTensor
    return (x - MEAN) / STD                                                             # Normalize


def cut(img: Image.Image, n: int):                                                      # extract_fashioniq_patch.py:18-44
    w, h = img.size
    iw, ih = int(w / n), int(h / n)
    return [img.crop((j * iw, i * ih, (j + 1) * iw, (i + 1) * ih)) for i in range(n) for j in range(n)]


def patch_images(img: Image.Image, dim: int = 224, target_ratio: float = 1.25) -> torch.Tensor:   # :142-156
    base = img.resize((360, 360), Image.LANCZOS)                                        # Image.ANTIALIAS
    return torch.stack([targetpad_transform(c, target_ratio, dim) for c in cut(base, 2) + cut(base, 3)])
