"""Composed-query pipeline: encode -> fuse -> rank for one batch, with several batches kept in flight.

One batch alone cannot keep 256 CUs busy through the whole step: the fusion stage is a chain of M = 64 GEMMs
(latency-bound, a few dozen workgroups each), every GEMM has a tile-quantisation tail, and the top-K merge is tiny.
The batches of a query stream are independent, so consecutive batches are dealt round-robin to `lanes` HIP streams,
each with its own forked native context (shared weights, private workspace); the hardware then fills one lane's
low-occupancy kernels with another lane's encoder tiles.  Results are bit-identical to the single-stream path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .engine import FernEngine


class QueryResult:
    """Top-K of one submitted batch; `scores` / `idx` are valid on the caller's stream after `wait()`."""

    def __init__(self, scores: torch.Tensor, idx: torch.Tensor, fused: torch.Tensor, event: torch.cuda.Event, member_scores=None,
                 start_event: Optional[torch.cuda.Event] = None):
        self.scores, self.idx, self.fused, self._event, self.member_scores = scores, idx, fused, event, member_scores
        self._start = start_event

    @property
    def start_event(self) -> Optional[torch.cuda.Event]:
        """``timing=True`` pipelines: recorded on the batch's lane in front of its first kernel, i.e. when the batch reaches the head
        of its lane -- `start_event.elapsed_time(done_event)` is the batch's service latency with the other lanes' batches in flight."""
        return self._start

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


class FusedResult:
    """Fused query features of one `submit_fuse` batch; valid on the caller's stream after `wait()`."""

    def __init__(self, fused: torch.Tensor, event: torch.cuda.Event, keep):
        self.fused, self._event, self._keep = fused, event, keep

    def wait(self) -> torch.Tensor:
        cur = torch.cuda.current_stream()
        cur.wait_event(self._event)
        self.fused.record_stream(cur)
        return self.fused

    def release(self) -> None:
        """Drop the host staging buffers of the batch once its uploads have run (the lane's event has completed): a long query loop
        must not hold every batch's pinned memory until its end (gigabytes at 200k queries)."""
        if self._keep is not None:
            self._event.synchronize()
            self._keep = None


class _LaneGraph:
    """One captured step of one lane: static input buffers, the hipGraph, the tensors it writes."""

    def __init__(self):
        self.calls = 0
        self.graph = None
        self.inputs = None
        self.outputs = None
        self.ws_generation = None      # the lane engine's workspace generation the graph's baked-in addresses belong to


class ComposedQueryPipeline:
    """``graphs=True``: after two eager calls with the same shapes (workspaces sized, tiles tuned) a lane's whole step -- about
    350 kernel launches at ViT-B/16 -- is captured once into a hipGraph and replayed: inputs are copied into the lane's static
    buffers, outputs are cloned out of the graph's, so the caller sees the same interface and bit-identical results, while the
    host enqueues one graph instead of hundreds of kernels and the kernels of a lane follow each other without launch gaps.
    A graph holds addresses inside its lane engine's workspace; when that engine later re-allocates the workspace (an eager
    call with a bigger batch / gallery / K, another precision, a direct call on ``engines[0]``) its `ws_generation` moves and
    every graph of the lane is dropped and re-captured on its next use instead of being replayed against freed memory."""

    _live: dict = {}      # id(pipeline) -> lanes of every pipeline that has not been closed (the tuner's concurrency is process-wide)

    def __init__(self, engine: FernEngine, lanes: int = 3, timing: bool = False, graphs: bool = False):
        self.timing = bool(timing)
        self.graphs = bool(graphs)
        self._lane_graphs = [dict() for _ in range(lanes)]
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        if engine.clip_cfg is None or engine.feature_dim is None:
            raise RuntimeError("the engine needs finalised CLIP and fusion weights")
        # process-wide tuner setting: shapes tuned from here on are scored for the most batches any LIVE pipeline keeps in flight
        ComposedQueryPipeline._live[id(self)] = lanes
        engine.tuner_set_concurrency(max(ComposedQueryPipeline._live.values()))
        self.engines: List[FernEngine] = [engine] + [engine.fork() for _ in range(lanes - 1)]
        self.streams = [torch.cuda.Stream(device=engine.device) for _ in range(lanes)]
        self._next = 0
        self._seen_jobs: set = set()

    def _runs_alone(self, key) -> bool:
        """True for the FIRST job of a (precision, shapes) key: the caller drains the lanes before it and waits for it, so that the GEMM
        tuner's one-off trials for the job's shapes (csrc/gemm.hip, gemm_bf16.hip: tile choice, image + text pair form) are timed with no
        other lane's kernels beside them -- four lanes meeting a new shape at once timed each other's trials (round 6: the pair form
        of c_proj flipped between runs).  One drain per key; every later job of the key is asynchronous."""
        key = (self.engines[0].precision,) + key
        if key in self._seen_jobs:
            return False
        self._seen_jobs.add(key)
        self.synchronize()
        return True

    def submit(self, images: Optional[torch.Tensor], tokens: torch.Tensor, local: torch.Tensor, gallery: torch.Tensor, k: int,
               exclude_idx=None, members=None, idx_offset: int = 0, ref_feats: Optional[torch.Tensor] = None) -> QueryResult:
        """images [B,3,S,S], tokens [B,77] int64, local [B,13,D] (device tensors), fused gallery [N,D] fp32 -- or a `PreparedGallery`
        of it (engine.prepare_gallery: the same exact fp32 ranking through the certified bf16 pre-filter, the form a serving process
        keeps), or bf16, which selects the bf16 sweep -- -> QueryResult.  `exclude_idx` [B] drops one gallery index per query and `members` [B,m]
        (fp32 gallery) also returns the scores of those rows: CIRR's reference removal and subset ranking
        (run/test/test_cirr.py:55-66).  `ref_feats` [B,D] (with `images=None`) is the reference harness's own query form: the
        reference image's RAW feature is looked up in the gallery index instead of being encoded again (test_fiq.py:104-107), so
        the step is text tower + fusion + rank."""
        if (images is None) == (ref_feats is None):
            raise ValueError("give either images (encoded per query) or ref_feats (looked up in the index), not both / neither")
        if members is not None and gallery.dtype != torch.float32:
            raise ValueError("members (subset scores) need an fp32 gallery: gather_scores has no bf16 form, and converting the "
                             "gallery per step would copy all of it")
        alone = self._runs_alone(("rank", None if images is None else tuple(images.shape), tuple(tokens.shape), tuple(gallery.shape), gallery.dtype))
        lane = self._next
        self._next = (self._next + 1) % len(self.engines)
        eng, stream = self.engines[lane], self.streams[lane]
        stream.wait_stream(torch.cuda.current_stream())          # inputs produced on the caller's stream
        args = (images, tokens, local, exclude_idx, members, ref_feats)
        with torch.cuda.stream(stream):
            ev0 = None
            if self.timing:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record(stream)
            if self.graphs:
                outs = self._replay(lane, eng, stream, args, gallery, k, idx_offset)
            else:
                outs = self._step(eng, args, gallery, k, idx_offset)
            ev = torch.cuda.Event(enable_timing=self.timing)
            ev.record(stream)
        if alone:
            stream.synchronize()
        fused, scores, idx, member_scores = outs
        return QueryResult(scores, idx, fused, ev, member_scores, ev0)

    def submit_fuse(self, tokens: torch.Tensor, local: torch.Tensor, ref_rows: torch.Tensor, index_features: torch.Tensor) -> "FusedResult":
        """The query loop of the reference harness as ONE lane job (run/test/test_fiq.py:98-118): tokens [B,77] int64 and local
        [B,13,D] as the DataLoader / tokenizer leave them -- HOST tensors (pinned ones upload asynchronously) or device tensors --,
        `ref_rows` [B] int64 = the gallery rows of the reference images, whose RAW features are looked up in `index_features`
        (test_fiq.py:104-107); text tower (one pass for global + seq) + `mode="test"` fusion -> fused [B,D].  Uploads, lookup and
        kernels all go to the lane's stream: batch i + 1 is tokenised / uploaded by the host while batch i's kernels run, and
        consecutive batches overlap on the lanes.  No synchronisation; `FusedResult.wait()` orders the caller's stream behind it."""
        alone = self._runs_alone(("fuse", tuple(tokens.shape)))
        lane = self._next
        self._next = (self._next + 1) % len(self.engines)
        eng, stream = self.engines[lane], self.streams[lane]
        stream.wait_stream(torch.cuda.current_stream())
        keep = (tokens, local, ref_rows)                              # host staging buffers must outlive their asynchronous copies
        with torch.cuda.stream(stream):
            dev = eng.device
            tk = tokens.to(dev, non_blocking=True)
            lc = local.to(dev, dtype=torch.float32, non_blocking=True)
            rows = ref_rows.to(index_features.device, non_blocking=True)
            ref = index_features[rows].to(dev)
            tg, ts = eng.encode_text(tk)
            fused = eng.dvr_fuse(ref, lc, tg, ts)
            ev = torch.cuda.Event()
            ev.record(stream)
        if alone:
            stream.synchronize()
        return FusedResult(fused, ev, keep)

    @staticmethod
    def _step(eng, args, gallery, k, idx_offset):
        images, tokens, local, exclude_idx, members, ref_feats = args
        if ref_feats is None and images.shape[0] == tokens.shape[0]:
            ref, tg, ts = eng.encode_pair(images, tokens)      # both towers in one pass: the text layers' GEMMs ride in the image layers' launches (fp32 / f32x3 / mx8img)
        else:
            ref = eng.encode_image(images) if ref_feats is None else ref_feats
            tg, ts = eng.encode_text(tokens)
        fused = eng.dvr_fuse(ref, local, tg, ts)
        if gallery.dtype == torch.bfloat16:
            scores, idx = eng.sim_topk_bf16(fused, gallery, k, idx_offset=idx_offset, exclude_idx=exclude_idx)
        else:
            scores, idx = eng.sim_topk(fused, gallery, k, idx_offset=idx_offset, exclude_idx=exclude_idx)
        member_scores = eng.gather_scores(fused, gallery, members) if members is not None else None
        return fused, scores, idx, member_scores

    def _replay(self, lane, eng, stream, args, gallery, k, idx_offset):
        key = tuple((tuple(a.shape), a.dtype) if a is not None else None for a in args) + (
            gallery.data_ptr(), tuple(gallery.shape), gallery.dtype, int(k), int(idx_offset), eng.precision)
        lg = self._lane_graphs[lane].setdefault(key, _LaneGraph())
        if lg.graph is not None and lg.ws_generation != eng.ws_generation():
            stream.synchronize()                                 # the workspace the graph points into was freed: start over
            lg.graph, lg.outputs, lg.calls = None, None, 0
        lg.calls += 1
        if lg.graph is None:
            if lg.calls <= 2:                                    # eager: sizes the workspaces, lets the tile tuner see every shape
                return self._step(eng, args, gallery, k, idx_offset)
            lg.inputs = tuple(None if a is None else a.clone() for a in args)
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                lg.outputs = self._step(eng, lg.inputs, gallery, k, idx_offset)
            lg.graph, lg.ws_generation = graph, eng.ws_generation()
        for dst, src in zip(lg.inputs, args):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        lg.graph.replay()
        return tuple(None if o is None else o.clone() for o in lg.outputs)

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
        self._lane_graphs = [dict() for _ in self._lane_graphs]
        for e in self.engines[1:]:
            e.close()
        self.engines = self.engines[:1]
        # process-wide setting: the survivors' maximum (a c2 and a c5 pipeline in one serving process: closing one must not make the
        # other's later shapes be scored for stand-alone latency -- ADVICE r4); stand-alone again only when the last one closes
        ComposedQueryPipeline._live.pop(id(self), None)
        self.engines[0].tuner_set_concurrency(max(ComposedQueryPipeline._live.values(), default=1))
