"""Shared body of the per-dataset harness modules.

The reference repeats the same ~100 lines per dataset (run/test/test_*.py, run/valid/validate_*.py); the
arithmetic that matters is identical and lives here once:

* query loop  (test_fiq.py:67-122): caption formatting, tokenise, encode_text x2, RAW reference-feature lookup,
  ``model(..., mode="test")``;
* rank step   (test_fiq.py:45-50): normalise gallery -> ``mode="index"`` -> cosine ranking.  The reference
  materialises ``1 - Q @ G.T`` and fully argsorts it; only ranks < 50 (51 with the CIRR reference removed) and the
  scores of <= 6 named members are consumed, so this asks the engine for top-K (+ gathered member scores) instead;
* recall      (test_fiq.py:54-60, test_cirr.py:55-80, test_200k.py:53-60): name compares on the host, percentages
  computed exactly as the reference does (float32 ``sum / len`` then ``* 100``).

Under ``torch.distributed`` (torchrun, one process per GPU) the same functions shard their work and keep their
contracts on EVERY rank: the query loop runs on this rank's contiguous slice of the relative dataset and the fused queries are
gathered (so ``generate_*_val_predictions`` still returns all Q predictions in dataset order), the gallery is fused shard by
shard + one all-gather (``distributed.build_gallery``), each rank ranks its own slice of the queries against the replicated
gallery (no per-batch collective) and the [Q, K] index rows are gathered for the recall arithmetic.  With one process nothing
changes.
"""
from __future__ import annotations

from collections import Counter
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from .. import distributed as fd
from ..tokenizer import get_tokenizer
from ..utils import collate_fn, host_threads, make_loader

TOPK = 50


def _engine_of(model):
    eng = getattr(model, "engine", None)
    if eng is None:
        raise RuntimeError("the ranking step needs model.engine (a FernEngine): no CPU ranking fallback exists")
    return eng


def format_fiq_captions(captions) -> List[str]:
    """test_fiq.py:93-97: collated [2][B] caption lists -> 'Cap1 and cap2' per item."""
    flat = np.array(captions).T.flatten().tolist()
    return [f"{flat[i].strip('.?, ').capitalize()} and {flat[i + 1].strip('.?, ')}" for i in range(0, len(flat), 2)]


def generate_predictions(kind: str, clip_model, relative_val_dataset, model, index_names, index_features, device,
                         feature_dim, batch_size, num_workers, clip_model_name) -> Dict[str, object]:
    tokenizer = get_tokenizer(clip_model_name)
    device = torch.device(device)
    rank, world = fd.world_info()
    if world > 1:       # query data parallel: this rank's contiguous slice of the queries
        q_start, q_stop, _ = fd.shard_rows(len(relative_val_dataset), rank, world)
        relative_val_dataset = Subset(relative_val_dataset, range(q_start, q_stop))
    loader = make_loader(relative_val_dataset, batch_size, num_workers, device, collate_fn)
    name_to_row = {n: i for i, n in enumerate(index_names)}      # duplicates: last row wins, like dict(zip(...)) (:88)
    # The HIP encoder + fusion on one engine: the loop below keeps its reference shape (same batches, same arithmetic, bit-identical
    # features) but every batch is ONE asynchronous lane job of a ComposedQueryPipeline -- uploads, the RAW reference-feature
    # lookup, one text-tower pass (the reference's two encode_text calls) and mode="test" -- so the host tokenises batch i + 1 while
    # batch i's kernels run and consecutive batches overlap on three streams; the only wait is at the end of the loop.
    # FERN_HARNESS_LANES=0 keeps the plain call-by-call loop (any clip_model / model objects take that path anyway).
    pipe = _query_pipeline(clip_model, model, device)
    pipe_box = [pipe]
    pending = []
    predicted: List[torch.Tensor] = []
    target_names: List[str] = []
    reference_names: List[str] = []
    group_members: List[List[str]] = []
    def parse(batch):
        members = None
        if kind in ("fiq", "val"):
            ref_names, batch_targets, captions, ref_patch = batch
            captions = format_fiq_captions(captions)
        elif kind == "cirr":
            ref_names, batch_targets, captions, ref_patch, members = batch
            members = np.array(members).T.tolist()                                   # test_cirr.py:113
        elif kind == "shoes":
            ref_names, batch_targets, captions, ref_patch, _ = batch
        elif kind == "200k":
            _, ref_names, captions, batch_targets, _, ref_patch = batch              # test_200k.py:89
        else:
            raise ValueError(kind)
        return ref_names, batch_targets, captions, ref_patch, members

    def prepared_batches():
        """Fast path: everything the host does per batch BEFORE the engine sees it -- DataLoader iteration, caption formatting,
        tokenising, the name -> row lookup, pinning -- runs in a producer thread, a few batches ahead of the consumer (the main
        thread spends its time inside libfern calls, which release the GIL)."""
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=6)
        stop = threading.Event()

        def produce():
            try:
                for batch in loader:
                    ref_names, batch_targets, captions, ref_patch, members = parse(batch)
                    text_inputs = tokenizer(list(captions), context_length=77)
                    if not text_inputs.is_cuda and text_inputs.numel() and (int(text_inputs.min()) < 0 or int(text_inputs.max()) >= clip_model.cfg.vocab_size):
                        raise IndexError(f"token id out of range [0, {clip_model.cfg.vocab_size})")      # nn.Embedding's error in the reference
                    rows = torch.tensor([name_to_row[n] for n in ref_names], dtype=torch.int64)
                    if device.type == "cuda" and not text_inputs.is_cuda:
                        text_inputs, rows = text_inputs.pin_memory(), rows.pin_memory()
                    item = (ref_names, batch_targets, members, text_inputs, ref_patch, rows)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                last = None
            except BaseException as e:      # noqa: BLE001 -- handed to the consumer, which re-raises it
                last = e
            # the terminal item obeys the same stop / timeout protocol as the batches: a consumer that has gone away while the queue
            # is full must not leave this thread (and the loader, its worker pool and the pinned batches it holds) blocked forever
            while not stop.is_set():
                try:
                    q.put(last, timeout=0.1)
                    return
                except queue.Full:
                    continue

        t = threading.Thread(target=produce, name="fern-harness-producer", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()

    with host_threads():
        if pipe is not None:
            window = 4 * len(pipe.engines)      # results older than this are collected: their pinned host batches (B x 13 x D fp32) go back
            for ref_names, batch_targets, members, text_inputs, ref_patch, rows in prepared_batches():
                pending.append(_submit_fuse(pipe_box, clip_model, model, device, text_inputs, ref_patch, rows, index_features))
                if len(pending) - len(predicted) > window:
                    done = pending[len(predicted)]
                    predicted.append(done.wait())
                    done.release()
                target_names.extend(batch_targets)
                reference_names.extend(ref_names)
                if members is not None:
                    group_members.extend(members)
        else:
            for batch in loader:
                ref_names, batch_targets, captions, ref_patch, members = parse(batch)
                text_inputs = tokenizer(list(captions), context_length=77)
                ref_patch = ref_patch.to(device)
                with torch.no_grad():
                    visual_emb = ref_patch.transpose(0, 1)                                   # [13,B,D]  (:101)
                    text_features, _ = clip_model.encode_text(text_inputs, visual_emb=visual_emb)              # :102
                    text_seq = clip_model.encode_text(text_inputs, mode="seq", visual_emb=visual_emb)          # :103
                    rows = torch.as_tensor([name_to_row[n] for n in ref_names], device=index_features.device)
                    ref_feats = index_features[rows]                                         # RAW gallery features (:104-107)
                    fused = model(ref_feats=ref_feats.to(device), ref_local_feats=ref_patch, text_feats=text_features.to(device),
                                  text_seq_feats=text_seq.to(device), mode="test")           # :112-118
                predicted.append(fused)
                target_names.extend(batch_targets)
                reference_names.extend(ref_names)
                if members is not None:
                    group_members.extend(members)
    if pipe is not None:
        for p in pending[len(predicted):]:
            predicted.append(p.wait())
            p.release()
    pred = torch.cat(predicted, dim=0) if predicted else torch.empty((0, feature_dim), device=device)
    if world > 1:       # every rank returns all Q predictions in dataset order, like the single-process loop
        per_row = [(t, r, group_members[i] if group_members else None) for i, (t, r) in enumerate(zip(target_names, reference_names))]
        pred, per_row = fd.gather_ragged(pred, per_row)
        target_names, reference_names = [p[0] for p in per_row], [p[1] for p in per_row]
        group_members = [p[2] for p in per_row] if any(p[2] is not None for p in per_row) else []
    return {"predicted": pred, "targets": target_names, "references": reference_names, "members": group_members}


def _query_pipeline(clip_model, model, device):
    """A 4-lane ComposedQueryPipeline (FERN_HARNESS_LANES; round 6: 4 -- one hardware queue per lane with GPU_MAX_HW_QUEUES=8, _lib.py) when the
    encoder and the fusion model are the HIP ones on ONE engine on a GPU, else None."""
    import os
    from ..clip_model import FernCLIP
    from ..engine import FernEngine
    lanes = int(os.environ.get("FERN_HARNESS_LANES", "4"))
    eng = getattr(model, "engine", None)
    if lanes < 1 or not isinstance(clip_model, FernCLIP) or not isinstance(eng, FernEngine) or clip_model.engine is not eng or device.type != "cuda":
        return None
    from ..pipeline import ComposedQueryPipeline
    # kept on the engine across calls (the reference's drivers evaluate category after category, test_fiq.py:176-190): forked
    # contexts allocate their workspaces on first use, which is worth paying once
    pipe = getattr(eng, "_harness_pipe", None)
    if pipe is None or len(pipe.engines) != lanes:
        if pipe is not None:
            pipe.close()
        pipe = eng._harness_pipe = ComposedQueryPipeline(eng, lanes=lanes)
    # A fork copies the parent's precision when it is MADE and fern_set_precision changes only the context it is called on: a caller
    # that switched the engine's precision since the last harness call (FernCLIP.set_precision / engine.set_precision) would get two of
    # every three batches at the old one.  The parent is the statement of what the caller wants.
    want = eng.precision
    if any(e.precision != want for e in pipe.engines[1:]):
        pipe.set_precision(want)
    return pipe


def _submit_fuse(pipe_box, clip_model, model, device, *args):
    """submit_fuse, rebuilding the cached pipeline once if its forks went stale (the weights were re-loaded since it was made)."""
    from .._lib import FernError
    try:
        return pipe_box[0].submit_fuse(*args)
    except FernError as e:
        if "stale" not in str(e):
            raise
        eng = model.engine
        eng._harness_pipe.close()
        eng._harness_pipe = None
        pipe_box[0] = _query_pipeline(clip_model, model, device)
        return pipe_box[0].submit_fuse(*args)


def fuse_index(model, index_features, index_local_features, prepared: bool = False):
    """test_fiq.py:45-46.  N ranks: each fuses ceil(N/W) rows, ONE all-gather replicates the fused gallery.  ``prepared``: under
    torch.distributed the ranking form of the gallery (`PreparedGallery`) is gathered too -- every rank prepares only its shard --
    and returned instead of the tensor (what `_ranked` would otherwise build from the whole gallery on every rank)."""
    return fd.build_gallery(_engine_of(model), index_features, index_local_features, normalize_input=True, prepared=prepared)


def _my_rows(q: int):
    rank, world = fd.world_info()
    return fd.shard_rows(q, rank, world)


def _ranked(model, predicted, index_fused, k, exclude=None) -> np.ndarray:
    """Top-k gallery rows of every query as a host array [Q, k]: this rank ranks its slice of the queries, the index rows are
    gathered (rank order == query order).  The engine is synchronised before the result is read, so a ranking error the
    kernels can only flag (fern_sync) raises here instead of being counted as misses."""
    eng = _engine_of(model)
    q = predicted.shape[0]
    start, stop, per = _my_rows(q)
    ex = None if exclude is None else torch.as_tensor(exclude[start:stop], dtype=torch.int32)
    if stop > start:
        # the gallery is ranked against once per evaluation, like the reference builds its index once (test_fiq.py:45-46): the
        # prepared form (bf16 pre-filter copy + its certificate) lets the engine pick the cheapest exact form of the stage
        prepared = eng.prepare_gallery(index_fused) if hasattr(eng, "prepare_gallery") and torch.is_tensor(index_fused) else index_fused
        _, idx = eng.sim_topk(predicted[start:stop], prepared, k, exclude_idx=ex)
    else:
        idx = torch.empty((0, k), dtype=torch.int32, device=predicted.device)
    if hasattr(eng, "sync"):
        eng.sync()
    if fd.world_info()[1] > 1:
        block = torch.full((per, k), -1, dtype=torch.int32, device=idx.device)
        block[: stop - start] = idx
        idx = fd.all_gather_shards(block, q)
    return idx.cpu().numpy()


def _pct(count: int, total: int) -> float:
    return (torch.tensor(int(count)) / total).item() * 100          # float32 tensor, like test_fiq.py:59


def _unique_rows(index_names: Sequence[str], wanted: Sequence[str], what: str) -> np.ndarray:
    counts = Counter(index_names)
    row = {n: i for i, n in enumerate(index_names)}
    for n in wanted:
        # the reference asserts exactly one ground-truth hit per ranking (test_fiq.py:56)
        assert counts.get(n, 0) == 1, f"{what} {n!r} occurs {counts.get(n, 0)} times in the index (expected exactly once)"
    return np.array([row[n] for n in wanted], dtype=np.int64)


def recalls_unique(model, predicted, index_fused, index_names, target_names, ks):
    """FashionIQ / Shoes / VAL: R@k = % of queries whose (unique) target is ranked < k."""
    q = len(target_names)
    tgt = _unique_rows(index_names, target_names, "target")
    hit = _ranked(model, predicted, index_fused, max(ks)) == tgt[:, None]
    return tuple(_pct(hit[:, :k].sum(), q) for k in ks)


def recalls_anyhit(model, predicted, index_fused, index_names, target_names, ks):
    """Fashion200k: gallery names are caption ids with duplicates; a hit is ANY of the top-k rows (test_200k.py:59-60)."""
    q = len(target_names)
    idx = _ranked(model, predicted, index_fused, max(ks))
    names = np.array(index_names)[idx.clip(min=0)]
    hit = (names == np.array(target_names)[:, None]) & (idx >= 0)
    return tuple(_pct((hit[:, :k].sum(1) > 0).sum(), q) for k in ks)


def recalls_cirr(model, predicted, index_fused, index_names, reference_names, target_names, group_members):
    """CIRR: reference image removed from each ranking; global R@1/5/10/50 + subset R@1/2/3 (test_cirr.py:55-80)."""
    q = len(target_names)
    eng = _engine_of(model)
    tgt = _unique_rows(index_names, target_names, "target")
    ref = _unique_rows(index_names, reference_names, "reference")
    hit = _ranked(model, predicted, index_fused, TOPK, exclude=ref) == tgt[:, None]
    glob = tuple(_pct(hit[:, :k].sum(), q) for k in (1, 5, 10, 50))
    # subset: rank the target among the query's img_set members (reference excluded) by the same scores
    row = {n: i for i, n in enumerate(index_names)}
    width = max(len(m) for m in group_members)
    member_rows = np.full((q, width), -1, dtype=np.int64)
    for i, mem in enumerate(group_members):
        rows = [row[m] for m in mem if m in row and m != reference_names[i]]
        assert rows.count(tgt[i]) == 1, "target must appear exactly once among the group members (test_cirr.py:69)"
        member_rows[i, :len(rows)] = rows
    start, stop, per = _my_rows(q)
    sc = torch.zeros((per if fd.world_info()[1] > 1 else q, width), dtype=torch.float32, device=predicted.device)
    if stop > start:
        sc[: stop - start] = eng.gather_scores(predicted[start:stop], index_fused, torch.as_tensor(member_rows[start:stop], dtype=torch.int32))
    sc = fd.all_gather_shards(sc, q).cpu().numpy().astype(np.float64)
    ranks = np.empty(q, dtype=np.int64)
    for i in range(q):
        rows_i = np.where(member_rows[i] >= 0, member_rows[i], np.iinfo(np.int64).max)
        order = np.lexsort((rows_i, -sc[i]))                   # score desc, gallery index asc
        ranks[i] = int(np.where(member_rows[i][order] == tgt[i])[0][0])
    grp = tuple(_pct((ranks < k).sum(), q) for k in (1, 2, 3))
    return grp + glob                                          # (G@1,G@2,G@3,R@1,R@5,R@10,R@50) test_cirr.py:80
