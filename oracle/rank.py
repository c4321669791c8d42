"""Oracle: cosine ranking, top-K and the Recall@K arithmetic of the reference harness.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch CPU fp32 / numpy.

The reference ranks with ``distances = 1 - q @ g.T`` and a full ``torch.argsort``
(/root/reference/run/test/test_fiq.py:49-50); only ranks < 50 (51 for CIRR) and the ranks of
<= 6 named members are ever consumed.  Tie rule (the reference leaves ties to an unstable sort):
score descending, then gallery index ascending == ``argsort(distances, stable=True)``.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch


def cosine_topk(q: torch.Tensor, g: torch.Tensor, k: int, idx_offset: int = 0, exclude_idx=None):
    """Top-k of ``q @ g.T`` per query row: (scores [B,k] desc, idx [B,k] int32 = local row + idx_offset).

    ``exclude_idx`` ([B] int, global index or -1): that gallery row is removed from the query's ranking
    (CIRR drops the reference image, test_cirr.py:55-58).  Slots beyond the available rows get
    score -inf and index -1.
    """
    scores = (q.float() @ g.float().T)
    b, n = scores.shape
    if exclude_idx is not None:
        ex = torch.as_tensor(exclude_idx, dtype=torch.long) - idx_offset
        ok = (ex >= 0) & (ex < n)
        rows = torch.arange(b)[ok]
        scores = scores.clone()
        scores[rows, ex[ok]] = float("-inf")
    order = torch.sort(scores, dim=1, descending=True, stable=True)
    kk = min(k, n)
    out_s = torch.full((b, k), float("-inf"))
    out_i = torch.full((b, k), -1, dtype=torch.int32)
    out_s[:, :kk] = order.values[:, :kk]
    out_i[:, :kk] = (order.indices[:, :kk] + idx_offset).to(torch.int32)
    dead = torch.isinf(out_s) & (out_s < 0)
    out_i[dead] = -1
    return out_s, out_i


def gather_scores(q: torch.Tensor, g: torch.Tensor, idx: torch.Tensor):
    """scores[b, j] = q[b] . g[idx[b, j]] (idx < 0 -> -inf): CIRR subset ranking, test_cirr.py:64-66."""
    idx = idx.long()
    safe = idx.clamp(min=0)
    s = (q.float().unsqueeze(1) * g.float()[safe]).sum(-1)
    return torch.where(idx >= 0, s, torch.full_like(s, float("-inf")))


def topk_merge(scores: torch.Tensor, idx: torch.Tensor):
    """Merge R per-shard top-K lists [R,B,K] into the global top-K [B,K] (score desc, index asc)."""
    r, b, k = scores.shape
    s = scores.permute(1, 0, 2).reshape(b, r * k)
    i = idx.permute(1, 0, 2).reshape(b, r * k).long()
    out_s = torch.empty(b, k)
    out_i = torch.empty(b, k, dtype=torch.int32)
    for row in range(b):
        ii = i[row].clone()
        ii[ii < 0] = np.iinfo(np.int64).max
        order = np.lexsort((ii.numpy(), -s[row].numpy().astype(np.float64)))[:k]
        out_s[row] = s[row][order]
        out_i[row] = i[row][order].to(torch.int32)
    return out_s, out_i


# ----------------------------------------------------------------------------------------
# Recall@K exactly as the reference harness computes it (full sort + name compares)
# ----------------------------------------------------------------------------------------

def _sorted_names(pred: torch.Tensor, index_feats: torch.Tensor, index_names: Sequence[str]):
    distances = 1 - pred.float() @ index_feats.float().T                 # test_fiq.py:49
    order = torch.argsort(distances, dim=-1, stable=True)                # :50 (stable = our tie rule)
    return np.array(index_names)[order.numpy()]                          # :51


def _pct(hits: torch.Tensor, n: int) -> float:
    return (hits.sum() / n).item() * 100                                 # float32 percent, test_fiq.py:59


def recall_unique(pred, index_feats, index_names, target_names, ks=(10, 50)):
    """FashionIQ / Shoes / VAL: exactly one gallery name equals the target (test_fiq.py:54-60)."""
    names = _sorted_names(pred, index_feats, index_names)
    labels = torch.tensor(names == np.array(target_names)[:, None])
    assert torch.equal(labels.sum(-1).int(), torch.ones(len(target_names)).int())
    return tuple(_pct(labels[:, :k], len(labels)) for k in ks)


def recall_cirr(pred, index_feats, index_names, reference_names, target_names, group_members: List[List[str]]):
    """CIRR: drop the reference, global R@1/5/10/50 and subset R@1/2/3 (test_cirr.py:50-80)."""
    names = _sorted_names(pred, index_feats, index_names)
    q, n = names.shape
    keep = names != np.array(reference_names)[:, None]                   # :55-56
    names = names[keep].reshape(q, n - 1)                                # :57-58
    labels = torch.tensor(names == np.array(target_names)[:, None])      # :60-61
    gm = np.array(group_members)
    group_mask = (names[..., None] == gm[:, None, :]).sum(-1).astype(bool)   # :64-65
    group_labels = labels[torch.from_numpy(group_mask)].reshape(q, -1)       # :66
    assert torch.equal(labels.sum(-1).int(), torch.ones(q).int())
    assert torch.equal(group_labels.sum(-1).int(), torch.ones(q).int())
    g = tuple(_pct(group_labels[:, :k], q) for k in (1, 2, 3))
    r = tuple(_pct(labels[:, :k], q) for k in (1, 5, 10, 50))
    return g + r                                                         # :80 ordering


def recall_anyhit(pred, index_feats, index_names, target_names, ks=(10, 50)):
    """Fashion200k: names are caption ids with duplicates; hit = any of the top-k rows (test_200k.py:53-60)."""
    names = _sorted_names(pred, index_feats, index_names)
    labels = torch.tensor(names == np.array(target_names)[:, None])
    return tuple(_pct(torch.where(labels[:, :k].sum(1) > 0, 1.0, 0.0), len(labels)) for k in ks)
