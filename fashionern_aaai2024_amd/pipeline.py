"""Composed-query pipeline: encode -> fuse -> rank for one batch, with several batches kept in flight.

One batch alone cannot keep 256 CUs busy through the whole step: the fusion stage is a chain of M = 64 GEMMs
(latency-bound, a few dozen workgroups each), every GEMM has a tile-quantisation tail, and the top-K merge is tiny.
The batches of a query stream are independent, so consecutive batches are dealt round-robin to `lanes` HIP streams,
each with its own forked native context (shared weights, private workspace); the hardware then fills one lane's
low-occupancy kernels with another lane's encoder tiles.  Results are bit-identical to the single-stream path.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from .engine import FernEngine


class QueryResult:
    """Top-K of one submitted batch; `scores` / `idx` are valid on the caller's stream after `wait()`."""

    def __init__(self, scores: torch.Tensor, idx: torch.Tensor, fused: torch.Tensor, event: torch.cuda.Event):
        self.scores, self.idx, self.fused, self._event = scores, idx, fused, event

    def wait(self) -> Tuple[torch.Tensor, torch.Tensor]:
        cur = torch.cuda.current_stream()
        cur.wait_event(self._event)
        for t in (self.scores, self.idx, self.fused):
            t.record_stream(cur)
        return self.scores, self.idx


class ComposedQueryPipeline:
    def __init__(self, engine: FernEngine, lanes: int = 3):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        if engine.clip_cfg is None or engine.feature_dim is None:
            raise RuntimeError("the engine needs finalised CLIP and fusion weights")
        self.engines: List[FernEngine] = [engine] + [engine.fork() for _ in range(lanes - 1)]
        self.streams = [torch.cuda.Stream(device=engine.device) for _ in range(lanes)]
        self._next = 0

    def submit(self, images: torch.Tensor, tokens: torch.Tensor, local: torch.Tensor, gallery: torch.Tensor, k: int,
               exclude_idx=None) -> QueryResult:
        """images [B,3,S,S], tokens [B,77] int64, local [B,13,D] (device tensors), fused gallery [N,D] -> QueryResult."""
        lane = self._next
        self._next = (self._next + 1) % len(self.engines)
        eng, stream = self.engines[lane], self.streams[lane]
        stream.wait_stream(torch.cuda.current_stream())          # inputs produced on the caller's stream
        with torch.cuda.stream(stream):
            ref = eng.encode_image(images)
            tg, ts = eng.encode_text(tokens)
            fused = eng.dvr_fuse(ref, local, tg, ts)
            scores, idx = eng.sim_topk(fused, gallery, k, exclude_idx=exclude_idx)
            ev = torch.cuda.Event()
            ev.record(stream)
        return QueryResult(scores, idx, fused, ev)

    def set_precision(self, precision) -> None:
        """Encoder operand precision of every lane ("fp32" parity mode / "bf16" perf mode, FernEngine.set_precision)."""
        self.synchronize()
        for e in self.engines:
            e.set_precision(precision)

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()

    def close(self) -> None:
        self.synchronize()
        for e in self.engines[1:]:
            e.close()
        self.engines = self.engines[:1]
