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


class ThreadedLoader:
    """`DataLoader(dataset, batch_size, shuffle=False, num_workers=W, pin_memory=..., collate_fn=...)` with W worker THREADS instead
    of W forked processes: batches come out in order, each built (items fetched, collated, pinned) by one worker, up to 2 W batches
    ahead of the consumer.

    Why not the reference's forked workers (`num_workers=4`, utils/utils.py:55, run/test/test_fiq.py:79-81) next to this GPU path:
    forking a process that owns a ROCm context write-protects every page of it (copy-on-write), and from then on each kernel launch
    of the PARENT -- kernel arguments, queue packets and signals live in host memory the driver has pinned -- takes the
    fault-and-re-pin path: one `encode_image` call of 32 images went from 9 ms to 373 ms while forked workers were alive, the whole
    `extract_index_features` loop from 3 100 to 160 images/s (profiles/r05_harness_profile_fork_workers.txt).  Workers from a clean
    forkserver avoid that but cost 1-3 s of start-up per loader (2 048 queries are 0.16 s of work).  Threads cost nothing to start and
    what a dataset item does off the GPU -- file reads, PIL decode / resize, tensor copies -- releases the GIL for most of its time."""

    def __init__(self, dataset, batch_size: int, num_workers: int, collate_fn, pin_memory: bool):
        self.dataset, self.batch_size, self.num_workers, self.collate_fn, self.pin = dataset, int(batch_size), max(1, int(num_workers)), collate_fn, pin_memory

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _build(self, lo: int, hi: int):
        batch = self.collate_fn([self.dataset[i] for i in range(lo, hi)])
        if self.pin:
            from torch.utils.data._utils.pin_memory import pin_memory
            batch = pin_memory(batch)
        return batch

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        n = len(self.dataset)
        spans = [(lo, min(lo + self.batch_size, n)) for lo in range(0, n, self.batch_size)]
        with ThreadPoolExecutor(max_workers=self.num_workers, thread_name_prefix="fern-loader") as pool:
            pending: deque = deque()
            nxt = 0
            while nxt < len(spans) or pending:
                while nxt < len(spans) and len(pending) < 2 * self.num_workers:
                    pending.append(pool.submit(self._build, *spans[nxt]))
                    nxt += 1
                yield pending.popleft().result()


def make_loader(dataset, batch_size: int, num_workers: int, device, collate):
    """The reference's DataLoader call, with worker threads instead of forked worker processes when the consumer is the GPU path
    (ThreadedLoader has the reason)."""
    device = torch.device(device)
    if device.type == "cuda" and num_workers > 0:
        return ThreadedLoader(dataset, batch_size, num_workers, collate, pin_memory=True)
    return DataLoader(dataset=dataset, batch_size=batch_size, num_workers=num_workers, pin_memory=(device.type == "cuda"), collate_fn=collate,
                      shuffle=False)


def extract_index_features(dataset, clip_model, patch_num, device, feature_dim, batch_size: int = 32,
                           num_workers: int = 4) -> Tuple[torch.Tensor, List[str], torch.Tensor]:
    """Gallery encode loop: ``(index_whole_features [N,D] raw, index_names, index_local_features [N,P,D])``.

    Same positional signature as the reference (batch 32 / 4 workers are its hard-coded values, utils.py:55-56).
    Outputs are written into pre-sized buffers instead of the reference's ``vstack``-in-a-loop, which re-copies the
    whole index every batch."""
    n = len(dataset)
    device = torch.device(device)
    loader = make_loader(dataset, batch_size, num_workers, device, collate_fn)
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
