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

    def __init__(self, scores: torch.Tensor, idx: torch.Tensor, fused: torch.Tensor, event: torch.cuda.Event, member_scores=None):
        self.scores, self.idx, self.fused, self._event, self.member_scores = scores, idx, fused, event, member_scores

    @property
    def done_event(self) -> torch.cuda.Event:
        """Recorded on the batch's lane when its last kernel was queued (timing-enabled when the pipeline was built with
        ``timing=True``: gaps between consecutive batches' events are per-batch service times)."""
        return self._event

    def wait(self) -> Tuple[torch.Tensor, torch.Tensor]:
        cur = torch.cuda.current_stream()
        cur.wait_event(self._event)
        for t in (self.scores, self.idx, self.fused):
            t.record_stream(cur)
        return self.scores, self.idx


class ComposedQueryPipeline:
    def __init__(self, engine: FernEngine, lanes: int = 3, timing: bool = False):
        self.timing = bool(timing)
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        if engine.clip_cfg is None or engine.feature_dim is None:
            raise RuntimeError("the engine needs finalised CLIP and fusion weights")
        self.engines: List[FernEngine] = [engine] + [engine.fork() for _ in range(lanes - 1)]
        self.streams = [torch.cuda.Stream(device=engine.device) for _ in range(lanes)]
        self._next = 0

    def submit(self, images: torch.Tensor, tokens: torch.Tensor, local: torch.Tensor, gallery: torch.Tensor, k: int,
               exclude_idx=None, members=None, idx_offset: int = 0) -> QueryResult:
        """images [B,3,S,S], tokens [B,77] int64, local [B,13,D] (device tensors), fused gallery [N,D] fp32 -- or bf16, which
        selects the bf16 sweep -- -> QueryResult.  `exclude_idx` [B] drops one gallery index per query and `members` [B,m]
        (fp32 gallery) also returns the scores of those rows: CIRR's reference removal and subset ranking
        (run/test/test_cirr.py:55-66)."""
        lane = self._next
        self._next = (self._next + 1) % len(self.engines)
        eng, stream = self.engines[lane], self.streams[lane]
        stream.wait_stream(torch.cuda.current_stream())          # inputs produced on the caller's stream
        with torch.cuda.stream(stream):
            ref = eng.encode_image(images)
            tg, ts = eng.encode_text(tokens)
            fused = eng.dvr_fuse(ref, local, tg, ts)
            if gallery.dtype == torch.bfloat16:
                scores, idx = eng.sim_topk_bf16(fused, gallery, k, idx_offset=idx_offset, exclude_idx=exclude_idx)
            else:
                scores, idx = eng.sim_topk(fused, gallery, k, idx_offset=idx_offset, exclude_idx=exclude_idx)
            member_scores = eng.gather_scores(fused, gallery, members) if members is not None else None
            ev = torch.cuda.Event(enable_timing=self.timing)
            ev.record(stream)
        return QueryResult(scores, idx, fused, ev, member_scores)

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
