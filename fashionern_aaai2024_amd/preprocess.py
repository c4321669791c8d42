"""Image side of the path on the GPU: PIL-exact resize, the reference's CLIP preprocessing and the 13-patch local features.

Reference: /root/reference/dataloader/dataset.py:31-87 (`TargetPad`, `targetpad_transform`: TargetPad -> Resize(dim, BICUBIC)
-> CenterCrop(dim) -> RGB -> ToTensor -> Normalize) and /root/reference/utils/extract_fashioniq_patch.py:18-44,142-160
(`image.resize((360, 360), ANTIALIAS)`, `cut_image_4`, `cut_image_9`, per-crop preprocess + `encode_image` -> [13, D]).

The resampling arithmetic itself is third-party Pillow (`Image.resize`, src/libImaging/Resample.c), restated here: the
host computes each output pixel's tap window and fixed-point coefficients exactly as `precompute_coeffs` /
`normalize_coeffs_8bpc` do; libfern's kernels do the integer accumulation (fern_resample_u8_*).  Results are
bit-identical to PIL (tests compare against PIL itself).  Image decoding (JPEG/PNG -> uint8 HWC) stays with the caller.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from functools import lru_cache
from typing import List, Tuple

import numpy as np
import torch

from . import _lib
from .engine import FernEngine, _ptr, _stream

PRECISION_BITS = 22                    # Pillow: 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # dataset.py:86
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:       # Resample.c bicubic_filter, a = -0.5
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:       # Resample.c lanczos_filter (Image.ANTIALIAS == LANCZOS)
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}


@lru_cache(maxsize=256)
def pil_coeffs(in_size: int, out_size: int, flt: str) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the full-width box: (bounds [out,2] int32, coeffs [out,ksize] int32)."""
    fn, sup = _FILTERS[flt]
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = sup * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


_dev_cache = {}
_dev_cache_lock = threading.Lock()      # dataset transforms run on ThreadedLoader's worker threads (include/fern.h: the image entry points are thread-safe)


def _dev_coeffs(engine: FernEngine, in_size: int, out_size: int, flt: str):
    key = (engine.device, in_size, out_size, flt)
    with _dev_cache_lock:
        if key not in _dev_cache:
            b, k = pil_coeffs(in_size, out_size, flt)
            _dev_cache[key] = (torch.from_numpy(b).to(engine.device), torch.from_numpy(k).to(engine.device), k.shape[1])
        return _dev_cache[key]


def resize_u8(engine: FernEngine, img: torch.Tensor, out_w: int, out_h: int, flt: str = "bicubic", box=None) -> torch.Tensor:
    """PIL `img.crop(box).resize((out_w, out_h), flt)` for an HWC uint8 RGB device tensor; box = (left, upper, right, lower)."""
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_cuda:
        raise ValueError("img must be a uint8 [H,W,3] device tensor")
    img = img.contiguous()
    big_h, big_w = int(img.shape[0]), int(img.shape[1])
    x0, y0, x1, y1 = box if box is not None else (0, 0, big_w, big_h)
    w, h = x1 - x0, y1 - y0
    if w <= 0 or h <= 0 or x0 < 0 or y0 < 0 or x1 > big_w or y1 > big_h:
        raise ValueError("crop box outside the image")
    lib, hnd = engine.lib, engine._h
    src, src_ld, sx, sy, rows = img, big_w, x0, y0, h
    if out_w != w:                                         # horizontal pass first (Resample.c ImagingResampleInner)
        hb, hk, hks = _dev_coeffs(engine, w, out_w, flt)
        tmp = torch.empty((h, out_w, 3), dtype=torch.uint8, device=engine.device)
        _lib.check(lib.fern_resample_u8_horizontal(hnd, _ptr(src), src_ld, sx, sy, rows, _ptr(tmp), out_w, _ptr(hb), _ptr(hk), hks,
                                                   _stream()), "fern_resample_u8_horizontal")
        src, src_ld, sx, sy = tmp, out_w, 0, 0
    if out_h != h:
        vb, vk, vks = _dev_coeffs(engine, h, out_h, flt)
        out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=engine.device)
        _lib.check(lib.fern_resample_u8_vertical(hnd, _ptr(src), src_ld, sx, sy, out_w, _ptr(out), out_h, _ptr(vb), _ptr(vk), vks,
                                                 _stream()), "fern_resample_u8_vertical")
        return out
    if src is img:                                         # same size: PIL returns a copy of the cropped region
        return img[y0:y1, x0:x1].contiguous()
    return src


def to_normalized_chw(engine: FernEngine, imgs: torch.Tensor, out_h: int, out_w: int, x0: int = 0, y0: int = 0,
                      mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """ToTensor + Normalize of n stacked HWC uint8 images [n,H,W,3] (or one [H,W,3]) with a crop window -> [n,3,out_h,out_w] f32."""
    if imgs.dim() == 3:
        imgs = imgs.unsqueeze(0)
    imgs = imgs.contiguous()
    n, big_h, big_w = int(imgs.shape[0]), int(imgs.shape[1]), int(imgs.shape[2])
    if y0 + out_h > big_h or x0 + out_w > big_w:
        raise ValueError("crop window outside the image")
    out = torch.empty((n, 3, out_h, out_w), dtype=torch.float32, device=engine.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    _lib.check(engine.lib.fern_u8_to_normalized_chw(engine._h, _ptr(imgs), big_w, x0, y0, big_h * big_w * 3, _ptr(out), n, out_h, out_w,
                                                    m, s, _stream()), "fern_u8_to_normalized_chw")
    return out


def target_pad(img: torch.Tensor, target_ratio: float) -> torch.Tensor:
    """`TargetPad.__call__` (dataset.py:46-54): zero-pad the short side until max/min <= target_ratio."""
    h, w = int(img.shape[0]), int(img.shape[1])
    if max(w, h) / min(w, h) < target_ratio:
        return img
    scaled = max(w, h) / target_ratio
    hp, vp = max(int((scaled - w) / 2), 0), max(int((scaled - h) / 2), 0)
    out = torch.zeros((h + 2 * vp, w + 2 * hp, 3), dtype=torch.uint8, device=img.device)
    out[vp:vp + h, hp:hp + w] = img
    return out


def targetpad_transform(engine: FernEngine, img: torch.Tensor, target_ratio: float = 1.25, dim: int = 288) -> torch.Tensor:
    """dataset.py:73-87 on one HWC uint8 RGB device image -> [3, dim, dim] f32."""
    img = target_pad(img, target_ratio)
    h, w = int(img.shape[0]), int(img.shape[1])
    if w <= h:                                             # torchvision Resize(int): smaller edge -> dim, other int(dim * long / short)
        nw, nh = dim, int(dim * h / w)
    else:
        nw, nh = int(dim * w / h), dim
    r = resize_u8(engine, img, nw, nh, "bicubic")
    top, left = int(round((nh - dim) / 2.0)), int(round((nw - dim) / 2.0))          # CenterCrop
    return to_normalized_chw(engine, r, dim, dim, x0=left, y0=top)[0]


def cut_boxes(width: int, height: int, n: int) -> List[Tuple[int, int, int, int]]:
    """`cut_image_4` (n=2) / `cut_image_9` (n=3) boxes, row-major (extract_fashioniq_patch.py:18-44)."""
    iw, ih = int(width / n), int(height / n)
    return [(j * iw, i * ih, (j + 1) * iw, (i + 1) * ih) for i in range(n) for j in range(n)]


def patch_images(engine: FernEngine, img: torch.Tensor, dim: int = 224, target_ratio: float = 1.25) -> torch.Tensor:
    """The 13 preprocessed crops of extract_fashioniq_patch.py:142-156 as one batch [13, 3, dim, dim] f32."""
    base = resize_u8(engine, img, 360, 360, "lanczos")                                       # :143 (ANTIALIAS)
    crops = []
    for box in cut_boxes(360, 360, 2) + cut_boxes(360, 360, 3):                              # :146-148
        crop = base[box[1]:box[3], box[0]:box[2]]
        crops.append(targetpad_transform(engine, crop.contiguous(), target_ratio, dim))      # square crops: TargetPad is a no-op
    return torch.stack(crops)


def extract_patch_features(clip_model, img: torch.Tensor, dim: int = None) -> torch.Tensor:
    """[13, D] local features of one image: 2x2 + 3x3 crops -> preprocess -> `encode_image` (extract_fashioniq_patch.py:150-156)."""
    engine = clip_model.engine
    dim = dim if dim is not None else clip_model.cfg.image_size
    return clip_model.encode_image(patch_images(engine, img, dim))


def save_patch_features(path: str, feats: torch.Tensor) -> None:
    """The on-disk format the reference's datasets read back (`torch.load(patch_path)` -> f32 [13, D] on the CPU;
    written at extract_fashioniq_patch.py:157-168, read at dataloader/fashioniq.py:69-70)."""
    torch.save(feats.detach().float().cpu(), path)


def load_patch_features(path: str) -> torch.Tensor:
    t = torch.load(path, map_location="cpu")
    if t.dim() != 2 or t.shape[0] != 13:
        raise ValueError(f"{path}: expected a [13, D] tensor, got {tuple(t.shape)}")
    return t.float()


def gpu_preprocess(engine: FernEngine, target_ratio: float = 1.25, dim: int = 288):
    """``PIL.Image -> [3, dim, dim] f32`` (CPU tensor, so that the DataLoader can collate and pin it) running TargetPad /
    Resize(BICUBIC) / CenterCrop / ToTensor / Normalize on the GPU: the callable the dataset classes take as ``preprocess``
    (the reference passes ``targetpad_transform(target_ratio, dim)``, dataloader/dataset.py:73-87).  Image *decoding* stays
    with PIL on the host.  Use with ``num_workers=0`` (the HIP context lives in this process)."""
    def run(image):
        if image.mode == "RGB":
            arr = torch.from_numpy(np.array(image, dtype=np.uint8)).to(engine.device)
            return targetpad_transform(engine, arr, target_ratio, dim).cpu()
        # The reference pads and resizes in the image's NATIVE mode and converts to RGB afterwards (dataset.py:73-87), and PIL's
        # resize is mode dependent (RGBA / LA: premultiplied alpha; P / 1: forced NEAREST; a P-mode pad fill of 0 is palette
        # entry 0, not black).  The GPU resampler implements the 8-bit RGB arithmetic only, so such files -- rare: FashionIQ /
        # CIRR ship RGB PNGs -- take PIL itself for pad + resize, beside the decode it already does; crop / ToTensor /
        # Normalize still run on the GPU.
        return pil_native_mode_transform(engine, image, target_ratio, dim).cpu()
    return run


def pil_native_mode_transform(engine: FernEngine, image, target_ratio: float, dim: int) -> torch.Tensor:
    """dataset.py:73-87 for a non-RGB PIL image: TargetPad and Resize(dim, BICUBIC) by Pillow in the image's own mode, then
    RGB conversion, then CenterCrop + ToTensor + Normalize through ``fern_u8_to_normalized_chw``."""
    from PIL import Image
    w, h = image.size
    if max(w, h) / min(w, h) >= target_ratio:                                       # TargetPad.__call__ (dataset.py:46-54)
        scaled = max(w, h) / target_ratio
        hp, vp = max(int((scaled - w) / 2), 0), max(int((scaled - h) / 2), 0)
        padded = Image.new(image.mode, (w + 2 * hp, h + 2 * vp), 0)                 # F.pad(image, [hp, vp], 0, 'constant')
        if image.mode == "P":
            padded.putpalette(image.getpalette())
        padded.paste(image, (hp, vp))
        image = padded
    w, h = image.size
    size = (dim, int(dim * h / w)) if w <= h else (int(dim * w / h), dim)           # torchvision Resize(int)
    image = image.resize(size, Image.BICUBIC).convert("RGB")
    nw, nh = image.size
    top, left = int(round((nh - dim) / 2.0)), int(round((nw - dim) / 2.0))          # CenterCrop
    arr = torch.from_numpy(np.array(image, dtype=np.uint8)).to(engine.device)
    return to_normalized_chw(engine, arr, dim, dim, x0=left, y0=top)[0]
