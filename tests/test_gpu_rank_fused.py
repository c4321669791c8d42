"""The fused sweep + top-K (no [B, N] score matrix): sample bound -> filtered sweep -> candidate select, through the C ABI.

Reference semantics: `distances = 1 - q @ g.T; argsort(distances)[:, :K]` (run/test/test_fiq.py:49-50); the oracle
(oracle/rank.py) is pinned to the imported reference by tests/golden.  These cases aim at the parts the plain parity
tests do not reach: a sample that misses the good rows (list overflow -> the exact pass), floods of exact ties at the bound,
the excluded row inside the sample, the bf16 sweep's 64-query blocks, galleries whose good rows are a multiple of 256 rows
apart, and lists forced down to one entry: the stage has no capacity left anywhere, every case must equal the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import rank as orank

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _int_unit(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-1, 2, (n, d), generator=g).float() / 8.0


def sample_row(c, r):
    """Restatement (test side) of kernels.h:sample_row -- which gallery row sample column c reads."""
    if r <= 1:
        return c
    h = (c * 2654435761) & 0xFFFFFFFF
    h ^= h >> 15
    return c * r + h % r


def plan(n):
    """Restatement of api.hip:rank_plan -- (sample columns S, ratio R)."""
    s = max(n // 64, min(n, 4096))
    return s, (n // s if s else 1)


def test_sample_misses_every_good_row_overflow_then_exact_pass(engine):
    """Query 0 scores its lowest on exactly the sampled rows, so its bound is useless and ~N rows reach it: the candidate lists
    overflow, the select kernel hands the query to the exact pass (rank_exact_kernel: the gallery streamed through sorted wave
    lists, scores from the sweep's own MFMA sequence).  The other queries take the normal path in the same launch.  (A query's
    candidates live in 256 lists of 64 entries: ~96k survivors are ~375 per list.)"""
    n, d, k = 100_000, 64, 50
    q, g = _int_unit(4, d, 21), _int_unit(n, d, 22)
    q[0] = torch.where(q[0] == 0, torch.full_like(q[0], 0.125), q[0])
    s, r = plan(n)
    assert r == 24 and (n - s) // 256 > 64
    rows = torch.tensor([sample_row(c, r) for c in range(s)])
    g[rows] = -torch.sign(q[0]) / 8.0                     # the worst possible score for query 0
    rs, ri = orank.cosine_topk(q, g, k)
    sc, ix = engine.sim_topk(q, g, k)
    assert torch.equal(ix.cpu(), ri) and torch.equal(sc.cpu(), rs)
    # the bf16 sweep shares the plan / select kernels (operands here are exact in bf16)
    sc, ix = engine.sim_topk_bf16(q, engine.gallery_to_bf16(g), k)
    engine.sync()
    assert torch.equal(ix.cpu(), ri) and torch.equal(sc.cpu(), rs)


@pytest.mark.parametrize("n", [1, 63, 1000, 1024, 1025, 2047, 4096, 65_537])
def test_all_scores_equal_only_the_index_breaks_ties(engine, n):
    """Every gallery row is the same vector: all scores tie, the ranking is 0, 1, 2, ... and the bound has to cut on the
    index part of the key or every row would survive."""
    d, k = 64, 50
    q = _int_unit(3, d, 5)
    g = _int_unit(1, d, 6).repeat(n, 1)
    rs, ri = orank.cosine_topk(q, g, k)
    sc, ix = engine.sim_topk(q, g, k)
    engine.sync()
    assert torch.equal(ix.cpu(), ri) and torch.equal(sc.cpu(), rs)
    assert ix[0, :min(n, k)].tolist() == list(range(min(n, k)))


def test_excluded_row_inside_the_sample_does_not_tighten_the_bound(engine):
    """CIRR reference removal: the excluded row is the query's best match AND a sampled row.  If it counted towards the K
    rows of the bound, the true K-th result could be filtered out."""
    n, d, k = 70_000, 64, 5
    s, r = plan(n)
    q, g = _int_unit(6, d, 31), _int_unit(n, d, 32)
    ex_rows = [sample_row(c, r) for c in (0, 17, 500, s - 1, 3, 9)]
    for b, row in enumerate(ex_rows):
        g[row] = torch.sign(q[b]) / 8.0 + (q[b] == 0) * 0.125     # the best possible score for query b
    ex = torch.tensor(ex_rows, dtype=torch.int32) + 7000
    rs, ri = orank.cosine_topk(q, g, k, idx_offset=7000, exclude_idx=ex)
    sc, ix = engine.sim_topk(q, g, k, idx_offset=7000, exclude_idx=ex)
    assert torch.equal(ix.cpu(), ri) and torch.equal(sc.cpu(), rs)
    assert not (ix.cpu() == ex[:, None]).any()
    sc, ix = engine.sim_topk_bf16(q, engine.gallery_to_bf16(g), k, idx_offset=7000, exclude_idx=ex)
    assert torch.equal(ix.cpu(), ri) and torch.equal(sc.cpu(), rs)


@pytest.mark.parametrize("b", [1, 63, 64, 65, 200])
def test_bf16_sweep_query_blocks_and_ragged_batches(engine, b):
    """fern_sim_topk_bf16 runs one sweep launch per 64-query block over shared plan buffers."""
    n, d, k = 30_011, 128, 50
    q, g = _int_unit(b, d, 41), _int_unit(n, d, 42)
    rs, ri = orank.cosine_topk(q, g, k)
    sc, ix = engine.sim_topk_bf16(q, engine.gallery_to_bf16(g), k)
    assert torch.equal(ix.cpu(), ri) and torch.equal(sc.cpu(), rs)


def test_large_query_batches_cross_the_plan_chunk(engine):
    """B > 1024 crosses the per-plan query chunk; B = 1024 / K = 51 with exclusions is BASELINE config C4's shape."""
    n, d = 21_552, 64
    q, g = _int_unit(1300, d, 51), _int_unit(n, d, 52)
    ex = torch.randint(0, n, (1300,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
    rs, ri = orank.cosine_topk(q, g, 51, exclude_idx=ex)
    sc, ix = engine.sim_topk(q, g, 51, exclude_idx=ex)
    engine.sync()
    assert torch.equal(ix.cpu(), ri) and torch.equal(sc.cpu(), rs)


def test_random_unit_rows_identical_order_up_to_fp32_near_ties(engine):
    """Random unit rows at the C3 shard size (D = 640): same top-50 as the oracle; a swap is tolerated only between
    neighbours the oracle's own fp32 scores separate by less than 2e-6."""
    from fashionern_aaai2024_amd import synth
    q, g = torch.from_numpy(synth.unit_rows(64, 640, tag="q3")), torch.from_numpy(synth.unit_rows(25_000, 640, tag="g3"))
    rs, ri = orank.cosine_topk(q, g, 50)
    sc, ix = engine.sim_topk(q, g, 50)
    sc, ix = sc.cpu(), ix.cpu()
    assert (sc - rs).abs().max().item() < 1e-5
    full = q.double() @ g.double().T
    for row, pos in (ix != ri).nonzero().tolist():
        assert abs(full[row, ix[row, pos]].item() - full[row, ri[row, pos]].item()) < 2e-6


_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from fashionern_aaai2024_amd.engine import FernEngine
from oracle import rank as orank
eng = FernEngine("cuda:0")
g = torch.Generator().manual_seed(1)
unit = lambda n, d: torch.randint(-1, 2, (n, d), generator=g).float() / 8.0
mode = {mode!r}
if mode == "cluster":
    # lists of 4 entries; rows 4096.. are outside the sample (N = 5000 -> the sample is rows 0..4095) and all of them beat it for
    # query 0: ~3.5 survivors per list -> overflow -> exact pass for query 0, the other queries stay on the list path
    q, gal = unit(16, 64), unit(5000, 64)
    q[0] = torch.where(q[0] == 0, torch.full_like(q[0], 0.125), q[0])
    gal[:4096] = torch.where(gal[:4096] * q[0] > 0, -gal[:4096], gal[:4096])       # sampled rows: never positive against query 0
    gal[4096:] = torch.sign(q[0]) / 8.0 * (torch.rand(904, 64, generator=g) < 0.9)  # the rest: strongly positive, all different
    k = 10
elif mode == "many":
    # 70 queries, tie-heavy scores, lists of ONE entry: (nearly) every query overflows -> the exact pass takes them 32 at a time as the
    # A rows of its MFMA tiles (three chunks: 32 + 32 + 6), ragged last gallery tile (N % 32 != 0)
    q, gal = unit(70, 64), unit(3001, 64)
    k = 50
else:
    # lists of ONE entry, all scores equal, K = 64: every list overflows for every query -- the case that used to end in NaN / -1
    # and an error at the next sync.  The exact pass must return rows 0..63 (ties -> lower index) with the common score.
    q, gal = unit(2, 64), unit(1, 64).repeat(200000, 1)
    k = 64
rs, ri = orank.cosine_topk(q, gal, k)
for ex in (None, torch.tensor([3] + [-1] * (q.shape[0] - 1), dtype=torch.int32)):
    if ex is not None:
        rs, ri = orank.cosine_topk(q, gal, k, exclude_idx=ex)
    s, i = eng.sim_topk(q, gal, k, exclude_idx=ex)
    eng.sync()
    assert torch.equal(i.cpu(), ri) and torch.equal(s.cpu(), rs), (mode, "fp32", ex is not None)
    s, i = eng.sim_topk_bf16(q, eng.gallery_to_bf16(gal), k, exclude_idx=ex)      # operands are exact in bf16: same oracle
    eng.sync()
    assert torch.equal(i.cpu(), ri) and torch.equal(s.cpu(), rs), (mode, "bf16", ex is not None)
s, i = eng.sim_topk(q, gal, k, idx_offset=1000)
assert torch.equal(i.cpu(), orank.cosine_topk(q, gal, k, idx_offset=1000)[1])
print("OK")
"""


@pytest.mark.parametrize("cap,mode", [("4", "cluster"), ("1", "all-equal"), ("1", "many")])
def test_forced_tiny_lists_go_through_the_exact_pass(cap, mode):
    """FERN_RANK_CAP (test hook, read once per process) shrinks the candidate lists so that they overflow; the result must
    still be the oracle's, bit for bit (the stage is exact by construction, no NaN rows, no late error)."""
    env = dict(os.environ, FERN_RANK_CAP=cap)
    r = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT, mode=mode)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_periodic_gallery_does_not_pile_into_one_list(engine):
    """a base set tiled up to a large gallery (period a multiple of 256) puts every copy of a top item 256k rows apart;
    with list = row % 256 they all shared one list and overflowed it twice.  Lists are now chosen by a hash of the row's 256-block,
    and whatever still overflows is ranked by the exact pass: the top-50 are the 50 lowest-index copies of the best items."""
    base = _int_unit(1024, 64, 31)
    g = base.repeat(300, 1)                                # 307 200 rows, period 1024
    q = _int_unit(5, 64, 32)
    rs, ri = orank.cosine_topk(q, g, 50)
    s, i = engine.sim_topk(q, g, 50)
    assert torch.equal(i.cpu(), ri) and torch.equal(s.cpu(), rs)
    s, i = engine.sim_topk_bf16(q, engine.gallery_to_bf16(g), 50)
    assert torch.equal(i.cpu(), ri) and torch.equal(s.cpu(), rs)


def test_stale_fork_bad_tokens_and_short_sequences_fail_loudly(engine):
    from fashionern_aaai2024_amd import synth
    from fashionern_aaai2024_amd._lib import FernError
    from fashionern_aaai2024_amd.clip_model import create_model
    from fashionern_aaai2024_amd.model import ERN
    cfg = synth.CLIP_CONFIGS["tiny"]
    d = cfg.embed_dim
    clip = create_model(cfg, device="cuda:0", seed=1)
    model = ERN(clip, d, "cuda:0", engine=clip.engine).init_random(2)
    eng = clip.engine
    fork = eng.fork()
    toks = torch.from_numpy(synth.captions(3, cfg))
    g0, s0 = fork.encode_text(toks.cuda())
    # re-finalising frees the weights the fork points at: the fork must refuse to run, a new fork works
    clip.init_random(1)
    with pytest.raises(FernError, match="stale"):
        fork.encode_text(toks.cuda())
    with pytest.raises(FernError, match="stale"):
        fork.encode_image(torch.from_numpy(synth.images(1, cfg)).cuda())
    g1, s1 = eng.fork().encode_text(toks.cuda())
    assert torch.equal(g0, g1) and torch.equal(s0, s1)
    # token ids outside the vocabulary: IndexError for host tokens (nn.Embedding's error), a reported flag + NaN for device tokens
    bad = toks.clone()
    bad[1, 4] = cfg.vocab_size + 3
    with pytest.raises(IndexError):
        eng.encode_text(bad)
    gb, _ = eng.encode_text(bad.cuda())
    with pytest.raises(FernError, match="token id outside"):
        eng.sync()
    assert torch.isnan(gb[1]).all() and torch.equal(gb[0], g1[0]) and torch.equal(gb[2], g1[2])
    eng.sync()
    # fewer than 13 text rows cannot feed BatchNorm1d(13) (fusion_model.py:47-48)
    loc = torch.from_numpy(synth.local_feats(3, d)).cuda()
    with pytest.raises(ValueError):
        model.engine.dvr_fuse(g1, loc, g1, s1[:, :12].contiguous())
    with pytest.raises(FernError, match="seq_len"):
        import ctypes as C
        out = torch.empty(3, d, device="cuda")
        from fashionern_aaai2024_amd import _lib
        _lib.check(eng.lib.fern_dvr_fuse(eng._h, C.c_void_p(g1.data_ptr()), C.c_void_p(loc.data_ptr()), C.c_void_p(g1.data_ptr()),
                                         C.c_void_p(s1.data_ptr()), C.c_void_p(out.data_ptr()), 3, 12, None), "fern_dvr_fuse")
    # visual_emb crosses the ABI: right shape accepted, wrong shape rejected by the library itself
    ve = loc.transpose(0, 1).contiguous()
    g2, _ = eng.encode_text(toks.cuda(), visual_emb=ve)
    assert torch.equal(g2, g1)
    with pytest.raises(FernError, match="visual_emb"):
        eng.encode_text(toks.cuda(), visual_emb=ve[:, :2].contiguous())
    eng.close()
