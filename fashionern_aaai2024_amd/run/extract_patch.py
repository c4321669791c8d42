"""``python -m fashionern_aaai2024_amd.run.extract_patch --images DIR --out DIR ...``: write the [13, D] local features the
dataset classes load (``<name>.pth`` next to each other), as /root/reference/utils/extract_fashioniq_patch.py:104-168 does for
Fashion200k: decode (PIL, host) -> resize 360x360 (ANTIALIAS) -> 2x2 + 3x3 crops -> TargetPad / bicubic / crop / normalise ->
``encode_image`` -> ``torch.save`` of a float32 [13, D] tensor.  Everything after the decode runs on the GPU
(``preprocess.extract_patch_features``).  Existing outputs are skipped (the reference keeps a ``dir.txt`` of finished paths)."""
from __future__ import annotations

import glob
import os
from argparse import ArgumentParser

import numpy as np
import torch

from ..clip_model import create_model
from ..preprocess import extract_patch_features, save_patch_features


def extract_directory(clip_model, image_paths, out_path_of, dim=None, overwrite=False, log_every=0) -> int:
    """Encode every image of ``image_paths`` that has no output yet; ``out_path_of(image_path) -> .pth path``.  Returns the number written."""
    import PIL.Image
    done = 0
    for i, path in enumerate(image_paths):
        out = out_path_of(path)
        if not overwrite and os.path.exists(out):
            continue
        with open(path, "rb") as f:
            img = torch.from_numpy(np.array(PIL.Image.open(f).convert("RGB"), dtype=np.uint8)).to(clip_model.device)
        feats = extract_patch_features(clip_model, img, dim)
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        save_patch_features(out, feats)
        done += 1
        if log_every and done % log_every == 0:
            print(f"{done} written ({i + 1}/{len(image_paths)} visited)", flush=True)
    return done


def main() -> None:
    p = ArgumentParser()
    p.add_argument("--images", required=True, help="directory that is searched recursively for images")
    p.add_argument("--out", required=True, help="output directory; the relative path of an image is kept, with '.pth' in place of the extension")
    p.add_argument("--pattern", default="*.png,*.jpg,*.jpeg", help="comma-separated glob patterns")
    p.add_argument("--keep-extension", action="store_true", help="append '.pth' to the full file name (the Fashion200k convention, x.jpeg.pth)")
    p.add_argument("--clip-model-name", default="ViT-B-16", type=str)
    p.add_argument("--clip-path", type=str, help="checkpoint with key 'CLIP' (open_clip state dict); default: seeded random init")
    p.add_argument("--seed", default=42, type=int)
    p.add_argument("--overwrite", action="store_true")
    args = p.parse_args()
    clip_model = create_model(args.clip_model_name, device=torch.device("cuda"), seed=None if args.clip_path else args.seed)
    if args.clip_path:
        clip_model.load_state_dict(torch.load(args.clip_path, map_location="cpu")["CLIP"])
    paths = sorted({q for pat in args.pattern.split(",") for q in glob.glob(os.path.join(args.images, "**", pat.strip()), recursive=True)})

    def out_path_of(path):
        rel = os.path.relpath(path, args.images)
        return os.path.join(args.out, (rel if args.keep_extension else os.path.splitext(rel)[0]) + ".pth")

    n = extract_directory(clip_model, paths, out_path_of, dim=clip_model.cfg.image_size, overwrite=args.overwrite, log_every=1000)
    print(f"{n} feature files written under {args.out} ({len(paths)} images found)")


if __name__ == "__main__":
    main()
