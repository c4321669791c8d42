"""Hot-path helpers of /root/reference/utils/utils.py: setup_seed (:15-19), collate_fn (:22-29),
extract_index_features (:44-69)."""
from __future__ import annotations

import random
from typing import List, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader


def setup_seed(seed: int) -> None:
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def collate_fn(batch: list):
    """Drop ``None`` items (datasets return None on a read error, fashioniq.py:104-105), then default-collate."""
    kept = [b for b in batch if b is not None]
    return torch.utils.data.dataloader.default_collate(kept)


def extract_index_features(dataset, clip_model, patch_num, device, feature_dim, batch_size: int = 32,
                           num_workers: int = 4) -> Tuple[torch.Tensor, List[str], torch.Tensor]:
    """Gallery encode loop: ``(index_whole_features [N,D] raw, index_names, index_local_features [N,P,D])``.

    Same positional signature as the reference (batch 32 / 4 workers are its hard-coded values, utils.py:55-56).
    Outputs are written into pre-sized buffers instead of the reference's ``vstack``-in-a-loop, which re-copies the
    whole index every batch."""
    n = len(dataset)
    device = torch.device(device)
    loader = DataLoader(dataset=dataset, batch_size=batch_size, num_workers=num_workers,
                        pin_memory=(device.type == "cuda"), collate_fn=collate_fn)
    whole = torch.empty((n, feature_dim), dtype=torch.float32, device=device)
    local = torch.empty((n, patch_num, feature_dim), dtype=torch.float32, device=device)
    names: List[str] = []
    at = 0
    for batch_names, images, local_feats in loader:
        images = images.to(device, non_blocking=True)
        with torch.no_grad():
            feats = clip_model.encode_image(images)
        b = feats.shape[0]
        whole[at:at + b] = feats.to(device)
        local[at:at + b] = local_feats.to(device, non_blocking=True)
        names.extend(batch_names)
        at += b
    return whole[:at], names, local[:at]       # at < n only if collate_fn dropped unreadable items


def concat_global_local_feats(global_feats: torch.Tensor, local_feats: torch.Tensor) -> torch.Tensor:
    """[B, D] and [B, P, D] -> [B, P + 1, D], global feature first (utils/utils.py:32-41)."""
    return torch.cat((global_feats.unsqueeze(1), local_feats), dim=1)


def _clean_caption(c: str) -> str:
    return c.strip(".?, ")


def generate_shoes_caption(flattened_captions: List[str]) -> List[str]:
    """One caption per query, trimmed of '.?, ' and capitalised (utils/utils.py:126-130)."""
    return [_clean_caption(c).capitalize() for c in flattened_captions]


def generate_randomized_fiq_caption(flattened_captions: List[str]) -> List[str]:
    """Training-time FashionIQ caption mixing (utils/utils.py:102-123): per pair, with probability 1/4 each, "A and b",
    "B and a", "A" or "B" (drawn with ``random.random()`` like the reference, so ``setup_seed`` makes it reproducible)."""
    import random
    out = []
    for i in range(0, len(flattened_captions), 2):
        a, b = _clean_caption(flattened_captions[i]), _clean_caption(flattened_captions[i + 1])
        u = random.random()
        if u < 0.25:
            out.append(f"{a.capitalize()} and {b}")
        elif 0.25 < u < 0.5:
            out.append(f"{b.capitalize()} and {a}")
        elif 0.5 < u < 0.75:
            out.append(a.capitalize())
        else:
            out.append(b.capitalize())
    return out
