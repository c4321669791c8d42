"""Multi-GPU layout of the path: one process per GPU, torch.distributed over RCCL ("nccl" on ROCm) / xGMI.

The reference's inference is single-GPU with no collective; sharding is new design:

* **gallery build** -- rank r fuses rows [r*ceil(N/W), ...) of the gallery (``mode="index"``), then ONE
  ``all_gather`` of the fused [N/W, D] blocks leaves the whole fused gallery on every GPU (46k x 512 fp32 = 94 MB;
  1M x 512 = 2 GB: trivially resident in 288 GB).  xGMI is point-to-point, so a single large all-gather (one shard
  per link) is the right shape -- there is nothing to bucket;
* **queries** -- data-parallel: every rank encodes, fuses and ranks its own B/W queries against the replicated
  gallery; no per-batch collective.  Results are gathered only for reporting;
* **sharded ranking** (alternative for very large N or tiny B): keep the gallery sharded, all-gather the fused
  queries [B, D], rank locally with ``idx_offset``, all-gather the [B, K] candidates and merge them
  (``topk_merge``: score desc, global index asc -- identical to the single-GPU ordering).

The compute object is an engine (``FernEngine`` in production).  Collectives go through ``torch.distributed`` so the
same code runs over RCCL on GPUs and over gloo in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank); a single process without those variables is (0, 1, 0) with no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # FERN_DIST_BACKEND overrides (debug: e.g. two ranks sharing the one GPU of a dev box over gloo)
            backend = os.environ.get("FERN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def world_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor) -> None:
    """`dist.all_gather_into_tensor(out, inp)` as BYTES (a bf16 / int32 block needs no dtype support from the backend), and
    through the host when the backend is gloo but the tensors live on a GPU (the debug layout: several ranks sharing the one GPU
    of a dev box, FERN_DIST_BACKEND=gloo) -- RCCL takes the device tensors as they are."""
    inp = inp.contiguous()
    if dist.get_backend() == "gloo" and inp.is_cuda:
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host.view(torch.uint8).view(-1), inp.cpu().view(torch.uint8).view(-1))
        out.copy_(host)
        return
    dist.all_gather_into_tensor(out.view(torch.uint8).view(-1), inp.view(torch.uint8).view(-1))


def shard_rows(n: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous equal shards of ceil(n / world) rows (the last ones may be short or empty): (start, stop, per)."""
    per = (n + world - 1) // world
    start = min(n, rank * per)
    return start, min(n, start + per), per


def build_gallery(engine, index_features: torch.Tensor, index_local: torch.Tensor, normalize_input: bool = True, prepared: bool = False):
    """Fused gallery [N, D] on every rank.  ``index_features`` / ``index_local`` are the FULL raw index (every rank
    holds or can address it); each rank fuses only its shard, then one all_gather replicates the result.
    ``prepared`` (world > 1, an engine with `prepare_gallery`): return the gathered `PreparedGallery` (`all_gather_prepared`: each rank
    prepares only its shard) instead of the fp32 tensor."""
    rank, world = world_info()
    n, d = index_features.shape
    if world == 1:
        return engine.index_fuse(index_features, index_local, normalize_input=normalize_input)
    start, stop, per = shard_rows(n, rank, world)
    dev = engine.device
    block = torch.zeros((per, d), dtype=torch.float32, device=dev)       # padded so every rank contributes `per` rows
    if stop > start:
        block[: stop - start] = engine.index_fuse(index_features[start:stop], index_local[start:stop],
                                                  normalize_input=normalize_input)
    if prepared and hasattr(engine, "prepare_gallery") and d % 4 == 0:
        return all_gather_prepared(engine, block, n)
    full = torch.empty((world * per, d), dtype=torch.float32, device=dev)
    _all_gather_into(full, block)
    return full[:n]


def all_gather_shards(block: torch.Tensor, n_total: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ONE all_gather of equally sized per-rank blocks [per, ...] (``per = ceil(n_total / world)``, short shards zero-padded by
    the caller) -> the first ``n_total`` rows of their concatenation on every rank.  This is the path's only data-path
    collective: one shard per xGMI link, nothing to bucket.  ``out`` ([world * per, ...], same dtype / device): the gallery
    store of a serving process, allocated when the process starts -- the collective writes straight into it (world 1: one copy)."""
    rank, world = world_info()
    if out is not None and (tuple(out.shape) != (world * block.shape[0],) + tuple(block.shape[1:]) or out.dtype != block.dtype
                            or out.device != block.device or not out.is_contiguous()):
        raise ValueError(f"out must be a contiguous [{world * block.shape[0]}, ...] {block.dtype} tensor on {block.device}")
    if world == 1:
        if out is None:
            return block[:n_total]
        out.copy_(block)
        return out[:n_total]
    block = block.contiguous()
    full = out if out is not None else torch.empty((world * block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    _all_gather_into(full, block)      # as bytes: a bf16 gallery (config 5) needs no bf16 support from the backend (gloo has none)
    return full[:n_total]


def _all_reduce_max(x: torch.Tensor) -> torch.Tensor:
    """In-place MAX all-reduce of a small fp32 tensor (through the host under the gloo debug backend with device tensors)."""
    if dist.get_backend() == "gloo" and x.is_cuda:
        h = x.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        x.copy_(h)
    else:
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return x


def all_gather_prepared(engine, block: torch.Tensor, n_total: int, out=None):
    """The replicated gallery in its PREPARED form (engine.prepare_gallery: fp32 rows + certified bf16 pre-filter copy + the three norms)
    without any rank preparing more than its own shard (round 5 all-gathered the fp32 blocks and then ran
    `fern_gallery_prepare` over the WHOLE gallery on every rank -- 35 ms at 1M rows, W times the work).  `block` [per, D] fp32 is this
    rank's fused shard, zero-padded to ``per = ceil(n_total / world)`` rows.  Each rank prepares its block; the fp32 blocks and the
    bf16 blocks are all-gathered (as bytes) and the norms MAX-reduced: a maximum of maxima is exact and a zero padding row has norm 0,
    so rows, bf16 copy and the four floats are bit for bit what preparing the gathered gallery gives.
    `out`: a `PreparedGallery` whose tensors are [world * per, D] stores to gather into (a serving process's gallery)."""
    from .engine import PreparedGallery
    rank, world = world_info()
    block = block.contiguous()
    mine = engine.prepare_gallery(block)
    if world == 1:
        if out is None:
            return PreparedGallery(mine.f32[:n_total], mine.bf16[:n_total], mine.meta)
        out.f32.copy_(mine.f32)
        out.bf16.copy_(mine.bf16)
        out.meta.copy_(mine.meta)
        return PreparedGallery(out.f32[:n_total], out.bf16[:n_total], out.meta)
    f32 = all_gather_shards(mine.f32, n_total, out=None if out is None else out.f32)
    b16 = all_gather_shards(mine.bf16, n_total, out=None if out is None else out.bf16)
    meta = mine.meta.clone() if out is None else out.meta.copy_(mine.meta)
    _all_reduce_max(meta)
    return PreparedGallery(f32, b16, meta)


def build_gallery_from_shard(engine, shard_features: torch.Tensor, shard_local: torch.Tensor, n_total: int,
                             normalize_input: bool = True) -> torch.Tensor:
    """Sharded gallery build: this rank holds ONLY its own rows [start, stop) of the raw index
    (``shard_rows(n_total, rank, world)``), fuses them with ``mode="index"`` and the fused blocks are all-gathered.  Unlike
    ``build_gallery`` no rank ever needs the whole raw index (6.6 GB of local features at C3, 27 GB at C5)."""
    rank, world = world_info()
    start, stop, per = shard_rows(n_total, rank, world)
    if shard_features.shape[0] != stop - start:
        raise ValueError(f"rank {rank} must hold rows [{start}, {stop}) of the index, got {shard_features.shape[0]} rows")
    block = torch.zeros((per, shard_features.shape[1]), dtype=torch.float32, device=engine.device)
    if stop > start:
        block[: stop - start] = engine.index_fuse(shard_features, shard_local, normalize_input=normalize_input)
    return all_gather_shards(block, n_total)


def rank_replicated(engine, queries: torch.Tensor, gallery: torch.Tensor, k: int, exclude_idx=None):
    """Query-data-parallel ranking: this rank's queries against the replicated gallery.  No collective."""
    return engine.sim_topk(queries, gallery, k, exclude_idx=exclude_idx)


def rank_sharded(engine, queries: torch.Tensor, gallery_shard: torch.Tensor, shard_start: int, k: int, exclude_idx=None):
    """Gallery-sharded ranking of the SAME query batch on every rank: local top-K with global indices, all-gather of
    the [B, K] candidates, merge.  Returns the global (scores, idx) on every rank."""
    rank, world = world_info()
    s, i = engine.sim_topk(queries, gallery_shard, k, idx_offset=shard_start, exclude_idx=exclude_idx)
    if world == 1:
        return s, i
    b, kk = s.shape
    all_s = torch.empty((world * b, kk), dtype=s.dtype, device=s.device)      # concatenated along dim 0 == [world, B, K]
    all_i = torch.empty((world * b, kk), dtype=i.dtype, device=i.device)
    _all_gather_into(all_s, s)
    _all_gather_into(all_i, i)
    return engine.topk_merge(all_s.view(world, b, kk), all_i.view(world, b, kk))


def share_gemm_tiles(engine, src: int = 0) -> None:
    """Every rank adopts rank `src`'s GEMM tile choices (the per-shape tuner runs independently in each process; all choices
    give bit-identical results, but different tiles run at slightly different speeds and a multi-GPU step is as slow as its
    slowest rank).  Call after the warm-up that visited the shapes."""
    rank, world = world_info()
    if world == 1:
        return
    box = [engine.tuner_export() if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    if rank != src:
        engine.tuner_import(box[0])


def gather_rows(x: torch.Tensor) -> torch.Tensor:
    """Concatenate equally-shaped per-rank results along dim 0 on every rank (reporting / recall on rank 0)."""
    rank, world = world_info()
    if world == 1:
        return x
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    _all_gather_into(out, x)
    return out


def gather_ragged(x: torch.Tensor, objs: Optional[list] = None):
    """Per-rank results with DIFFERENT row counts (a query shard whose loader dropped unreadable items, the short last shard)
    -> their concatenation in rank order on every rank.  `objs` (one Python object per row: names, member lists) travel with
    them.  Returns `x_all` or `(x_all, objs_all)`."""
    rank, world = world_info()
    if world == 1:
        return x if objs is None else (x, list(objs))
    meta = [None] * world
    dist.all_gather_object(meta, (int(x.shape[0]), None if objs is None else list(objs)))
    counts = [m[0] for m in meta]
    per = max(counts)
    block = torch.zeros((per,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    block[: x.shape[0]] = x
    full = torch.empty((world * per,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if per:
        _all_gather_into(full, block)
    out = torch.cat([full[r * per: r * per + c] for r, c in enumerate(counts)], dim=0)
    if objs is None:
        return out
    return out, [o for m in meta for o in m[1]]


def extract_index_features_sharded(dataset, clip_model, patch_num, device, feature_dim, batch_size: int = 32, num_workers: int = 0):
    """Gallery ENCODE sharded over the ranks (the other half of the gallery build): rank r runs the reference's
    `extract_index_features` loop (utils/utils.py:44-69) on items [r*ceil(N/W), ...) of `dataset`, then the raw features, the
    13-patch local features and the names are all-gathered so that every rank returns the same
    `(index_whole_features [N,D], index_names, index_local_features [N,P,D])` the single-GPU function returns."""
    from torch.utils.data import Subset

    from .utils import extract_index_features
    rank, world = world_info()
    n = len(dataset)
    if world == 1:
        return extract_index_features(dataset, clip_model, patch_num, device, feature_dim, batch_size, num_workers)
    start, stop, per = shard_rows(n, rank, world)
    device = torch.device(device)
    feats = torch.zeros((per, feature_dim), dtype=torch.float32, device=device)
    local = torch.zeros((per, patch_num, feature_dim), dtype=torch.float32, device=device)
    names = []
    if stop > start:
        f, names, l = extract_index_features(Subset(dataset, range(start, stop)), clip_model, patch_num, device, feature_dim,
                                             batch_size, num_workers)
        feats[: f.shape[0]] = f
        local[: l.shape[0]] = l
    all_f = torch.empty((world * per, feature_dim), dtype=torch.float32, device=device)
    all_l = torch.empty((world * per, patch_num, feature_dim), dtype=torch.float32, device=device)
    _all_gather_into(all_f, feats)
    _all_gather_into(all_l, local)
    gathered = [None] * world
    dist.all_gather_object(gathered, list(names))
    counts = [len(g) for g in gathered]
    keep = torch.cat([torch.arange(r * per, r * per + c, device=device) for r, c in enumerate(counts)]) if sum(counts) else \
        torch.empty(0, dtype=torch.long, device=device)
    return all_f[keep], [nm for g in gathered for nm in g], all_l[keep]
