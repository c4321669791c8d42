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


class host_threads:
    """Cap torch's intra-op thread pool for the duration of a host-side loop (restored on exit).  The loops of this harness do small
    host ops per batch -- collate / stack, pin, index -- and on a many-core host torch's default pool (one thread per logical core: 256
    on the benchmark box) spends ~100x the op itself forking and joining: `extract_index_features` ran at 143 images/s against an
    encoder doing 3 500.  The GPU path does not use the pool at all."""

    def __init__(self, cap: int = 16):
        self.cap = cap

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.cap:
            torch.set_num_threads(self.cap)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


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
    with host_threads():
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
