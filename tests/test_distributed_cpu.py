"""CPU, world_size 2 over gloo: the multi-GPU layout (gallery shard + all_gather, query DP, sharded top-K merge)
is placement-independent and reproduces the single-process ordering exactly.  Compute = TEST-ONLY OracleEngine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle_engine import OracleEngine
    from fashionern_aaai2024_amd import distributed as fd
    from fashionern_aaai2024_amd import synth
    r, w, _ = fd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    d = 128
    eng = OracleEngine()
    eng.load_tensors(synth.fusion_state_dict(d, seed=3))
    eng.finalize_fusion(d)
    raw, loc = torch.from_numpy(synth.global_feats(n, d, tag="g")), torch.from_numpy(synth.local_feats(n, d, tag="gl"))
    gallery = fd.build_gallery(eng, raw, loc)                                    # shard -> fuse -> all_gather
    q_all = torch.from_numpy(synth.unit_rows(10, d, tag="q"))
    q_mine = q_all[rank * 5:(rank + 1) * 5]                                      # query data parallel
    s, i = fd.rank_replicated(eng, q_mine, gallery, 7)
    s_all, i_all = fd.gather_rows(s), fd.gather_rows(i)
    start, stop, _ = fd.shard_rows(n, rank, world)
    # the sharded build proper: this rank holds ONLY its rows of the raw index (what bench.py does at C3 / C5 sizes)
    from_shard = fd.build_gallery_from_shard(eng, raw[start:stop], loc[start:stop], n)
    assert torch.equal(from_shard, gallery)
    # a bf16 gallery block (config 5) gathers the same way: bytes on the wire, no bf16 support needed from the backend
    _, _, per = fd.shard_rows(n, rank, world)
    blk = torch.zeros((per, d), dtype=torch.bfloat16)
    blk[: stop - start] = gallery[start:stop].bfloat16()
    assert torch.equal(fd.all_gather_shards(blk, n), gallery.bfloat16())
    # the PREPARED form of the replicated gallery: every rank prepares only its own shard, the fp32 and bf16 blocks
    # are all-gathered and the norms MAX-reduced -- rows, bf16 copy and the four floats equal those of preparing the whole gallery here
    pblk = torch.zeros((per, d))
    pblk[: stop - start] = gallery[start:stop]
    pg = fd.all_gather_prepared(eng, pblk, n)
    whole = eng.prepare_gallery(gallery)
    assert torch.equal(pg.f32, whole.f32) and torch.equal(pg.bf16.view(torch.int16), whole.bf16.view(torch.int16))
    assert torch.equal(pg.meta, whole.meta) and pg.meta[3] == 0 and tuple(pg.shape) == (n, d)
    store = eng.prepare_gallery(torch.zeros((world * per, d)))                  # a serving process's pre-allocated store
    pg2 = fd.all_gather_prepared(eng, pblk, n, out=store)
    assert torch.equal(pg2.f32, whole.f32) and torch.equal(pg2.bf16.view(torch.int16), whole.bf16.view(torch.int16)) and torch.equal(pg2.meta, whole.meta)
    assert pg2.f32.data_ptr() == store.f32.data_ptr() and pg2.bf16.data_ptr() == store.bf16.data_ptr()
    pgb = fd.build_gallery(eng, raw, loc, prepared=True)                        # what the harness's compute_* functions call
    assert torch.equal(pgb.f32, gallery) and torch.equal(pgb.meta, whole.meta)
    ps, pi = fd.rank_replicated(eng, q_mine, pgb, 7)
    assert torch.equal(pi, i) and torch.equal(ps, s)
    wrong = 0 if stop > start else 1                                             # any row count but the shard's own is refused
    with pytest.raises(ValueError):
        fd.build_gallery_from_shard(eng, raw[:wrong], loc[:wrong], n)
    ex = torch.tensor([3, -1, n - 1, 0, 5, -1, 7, 8, 9, 1], dtype=torch.int32)
    s_sh, i_sh = fd.rank_sharded(eng, q_all, gallery[start:stop], start, 7, exclude_idx=ex)
    # sharded gallery ENCODE: every rank ends up with the same (features, names, local features) as the single-process loop
    import synthetic_data as sdata
    gal = sdata.Gallery(n, d, seed=5)
    clip = sdata.StubCLIP(d).eval()
    ef, en, el = fd.extract_index_features_sharded(sdata.ClassicDataset(gal), clip, 13, "cpu", d, batch_size=16)
    if rank == 0:
        np.savez(os.path.join(out_dir, "r0.npz"), gallery=gallery.numpy(), s=s_all.numpy(), i=i_all.numpy(), s_sh=s_sh.numpy(), i_sh=i_sh.numpy(),
                 ef=ef.numpy(), el=el.numpy(), en=np.array(en))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [101, 64, 1])
def test_two_rank_layout_matches_single_process(tmp_path, n):
    sys.path.insert(0, HERE)
    from oracle_engine import OracleEngine
    from fashionern_aaai2024_amd import synth
    from oracle import rank as orank
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "r0.npz")
    d = 128
    eng = OracleEngine()
    eng.load_tensors(synth.fusion_state_dict(d, seed=3))
    raw, loc = torch.from_numpy(synth.global_feats(n, d, tag="g")), torch.from_numpy(synth.local_feats(n, d, tag="gl"))
    gallery = eng.index_fuse(raw, loc, normalize_input=True)
    assert np.abs(got["gallery"] - gallery.numpy()).max() < 1e-6       # CPU BLAS blocking differs with the shard size
    gallery = torch.from_numpy(got["gallery"])                         # ranking is checked bit-exactly on the gathered gallery
    q = torch.from_numpy(synth.unit_rows(10, d, tag="q"))
    s, i = orank.cosine_topk(q, gallery, 7)
    assert np.array_equal(got["i"], i.numpy()) and np.allclose(got["s"], s.numpy(), atol=1e-6)
    import synthetic_data as sdata
    from fashionern_aaai2024_amd.utils import extract_index_features
    gal = sdata.Gallery(n, d, seed=5)
    f1, n1, l1 = extract_index_features(sdata.ClassicDataset(gal), sdata.StubCLIP(d).eval(), 13, "cpu", d, num_workers=0)
    assert list(got["en"]) == n1 and np.allclose(got["ef"], f1.numpy(), atol=1e-6) and np.array_equal(got["el"], l1.numpy())
    ex = torch.tensor([3, -1, n - 1, 0, 5, -1, 7, 8, 9, 1], dtype=torch.int32)
    s2, i2 = orank.cosine_topk(q, gallery, 7, exclude_idx=ex)
    assert np.array_equal(got["i_sh"], i2.numpy()) and np.allclose(got["s_sh"], s2.numpy(), atol=1e-6)


def test_shard_rows_cover_everything_once():
    from fashionern_aaai2024_amd.distributed import shard_rows
    for n in (0, 1, 7, 8, 46000, 200001):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(stop - start <= per for start, stop, per in spans)


def _harness_worker(rank, world, port, out_dir):
    """The run/test_* harness under torch.distributed: sharded gallery encode, sharded gallery fusion + all-gather, query data
    parallel, gathered rankings.  Every rank must return the recall tuples of the single-process run."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import json
    import synthetic_data as sdata
    from oracle_engine import OracleEngine
    from fashionern_aaai2024_amd import distributed as fd
    from fashionern_aaai2024_amd import synth
    from fashionern_aaai2024_amd.model import ERN
    from fashionern_aaai2024_amd.run import test_200k, test_cirr, test_fiq, test_shoes, test_val
    from fashionern_aaai2024_amd.tokenizer import register_tokenizer
    fd.init_from_env("gloo")
    register_tokenizer("stub", sdata.stub_tokenizer)
    meta = json.load(open(os.path.join(HERE, "golden", "harness.json")))
    d, n, q = meta["d"], meta["n"], meta["q"]
    fns = {"fiq": test_fiq.compute_fiq_val_metrics, "cirr": test_cirr.compute_cirr_val_metrics, "200k": test_200k.compute_200k_val_metrics,
           "shoes": test_shoes.compute_shoes_val_metrics, "val": test_val.compute_fiq_val_metrics}
    out = {}
    for kind, fn in fns.items():
        clip = sdata.StubCLIP(d).eval()
        model = ERN(clip, d, "cpu", engine=OracleEngine())
        model.load_state_dict(synth.fusion_state_dict(d, seed=meta["fusion_seed"]))
        gal = sdata.Gallery(n, d, seed=meta["gallery_seed"], dup_names=(kind == "200k"))
        rel = sdata.RelativeDataset(gal, q, "fiq" if kind == "val" else kind, seed=meta["relative_seed"])
        feats, names, local = fd.extract_index_features_sharded(sdata.ClassicDataset(gal), clip, 13, "cpu", d, num_workers=0)
        out[kind] = list(fn(rel, clip, feats, local, names, model, "cpu", d, meta["batch_size"], 0, "stub"))
        if kind == "fiq":      # generate_* keeps its contract on every rank: all Q predictions, dataset order
            pred, targets = test_fiq.generate_fiq_val_predictions(clip, rel, model, names, feats, "cpu", d, meta["batch_size"], 0, "stub")
            assert targets == [it[1] for it in rel.items] and pred.shape == (q, d)
            out["fiq_pred"] = pred.numpy().tolist()
    json.dump(out, open(os.path.join(out_dir, f"harness_r{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_harness_under_torch_distributed_reproduces_reference_recalls(tmp_path, world):
    """SURVEY 8e / VERDICT r2 item 1c: run/test_{fiq,cirr,200k,shoes,val} with WORLD_SIZE > 1 (gloo) -- recall tuples equal the
    imported reference's (tests/golden/harness.json), on every rank; world 3 leaves ragged query / gallery shards."""
    import json
    port = _free_port()
    mp.spawn(_harness_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    meta = json.load(open(os.path.join(HERE, "golden", "harness.json")))
    gold = np.load(os.path.join(HERE, "golden", "harness.npz"))
    for r in range(world):
        got = json.load(open(tmp_path / f"harness_r{r}.json"))
        for kind in ("fiq", "cirr", "200k", "shoes", "val"):
            assert got[kind] == meta["recalls"][kind], (r, kind, got[kind], meta["recalls"][kind])
        assert np.abs(np.array(got["fiq_pred"], dtype=np.float32) - gold["fiq_predicted"]).max() < 1e-6


class _TunerStub:
    """The two tuner entry points `share_gemm_tiles` uses (engine.tuner_export / tuner_import), without a GPU."""

    def __init__(self, text):
        self.text = text

    def tuner_export(self):
        return self.text

    def tuner_import(self, text):
        self.text = text


def _world8_worker(rank, world, port, out_dir):
    """every collective helper of the path at WORLD = 8 over gloo with the block types the GPU job moves --
    bf16 gallery shards and int32 index blocks as BYTE views (no dtype support needed from the backend: the RCCL call sees uint8
    too), fp32 score blocks, ragged gathers with objects, the tuner-plan broadcast, and the sharded ranking's candidate gathers."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle_engine import OracleEngine
    from fashionern_aaai2024_amd import distributed as fd
    from fashionern_aaai2024_amd import synth
    r, w, _ = fd.init_from_env("gloo")
    assert (r, w) == (rank, world) and fd.world_info() == (rank, world)
    n, d, k = 1003, 64, 9                                              # ragged: the last shard is short
    start, stop, per = fd.shard_rows(n, rank, world)
    full = torch.from_numpy(synth.unit_rows(n, d, tag="w8"))
    # (1) bf16 shards through _all_gather_into / all_gather_shards, also into a pre-allocated store (`out=`, bench.py's gallery store)
    blk = torch.zeros((per, d), dtype=torch.bfloat16)
    blk[: stop - start] = full[start:stop].bfloat16()
    assert torch.equal(fd.all_gather_shards(blk, n), full.bfloat16())
    store = torch.empty((world * per, d), dtype=torch.bfloat16)
    view = fd.all_gather_shards(blk, n, out=store)
    assert view.data_ptr() == store.data_ptr() and torch.equal(view, full.bfloat16())
    with pytest.raises(ValueError):
        fd.all_gather_shards(blk, n, out=torch.empty((world * per, d), dtype=torch.float32))
    # (2) int32 and fp32 blocks through _all_gather_into directly (the [B, K] candidate blocks of the sharded ranking)
    idx = (torch.arange(5 * k, dtype=torch.int32).view(5, k) + 1000 * rank)
    out_i = torch.empty((world * 5, k), dtype=torch.int32)
    fd._all_gather_into(out_i, idx)
    assert torch.equal(out_i, torch.cat([torch.arange(5 * k, dtype=torch.int32).view(5, k) + 1000 * q for q in range(world)]))
    sc = torch.full((5, k), float(rank) + 0.5)
    out_s = torch.empty((world * 5, k))
    fd._all_gather_into(out_s, sc)
    assert torch.equal(out_s.view(world, 5, k)[:, 0, 0], torch.arange(world, dtype=torch.float32) + 0.5)
    # (3) gather_rows / gather_ragged (different row counts per rank, objects travel with the rows)
    assert torch.equal(fd.gather_rows(idx), out_i)
    mine = full[start:stop]
    names = [f"row{j}" for j in range(start, stop)]
    allx, allnames = fd.gather_ragged(mine, names)
    assert torch.equal(allx, full) and allnames == [f"row{j}" for j in range(n)]
    assert torch.equal(fd.gather_ragged(mine[: rank % 3]), torch.cat([full[fd.shard_rows(n, q, world)[0]:][: q % 3] for q in range(world)]))
    # (4) the tuner-plan broadcast: every rank adopts rank 0's text
    stub = _TunerStub(f"f32 12608 2304 768 0 0 {20 + rank % 2} 10880 12608\nmx8 12608 768 768 3 4 {rank}\n")
    fd.share_gemm_tiles(stub)
    assert stub.text == "f32 12608 2304 768 0 0 20 10880 12608\nmx8 12608 768 768 3 4 0\n"
    # (5) sharded ranking == replicated ranking (candidate all-gathers + merge at world 8), with an excluded row per query
    eng = OracleEngine()
    q = torch.from_numpy(synth.unit_rows(6, d, tag="w8q"))
    ex = torch.tensor([3, -1, n - 1, 0, 500, -1], dtype=torch.int32)
    s_sh, i_sh = fd.rank_sharded(eng, q, full[start:stop], start, k, exclude_idx=ex)
    s_re, i_re = fd.rank_replicated(eng, q, full, k, exclude_idx=ex)
    assert torch.equal(i_sh, i_re) and torch.equal(s_sh, s_re)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_world8_collectives_move_bf16_and_int32_blocks_as_bytes(tmp_path):
    port = _free_port()
    mp.spawn(_world8_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(8)]
