"""GPU: the drop-in surface end to end on the HIP engine -- ERN / FernCLIP / harness -- against (a) the recall tuples,
query features and top-50 lists captured from the imported reference harness (tests/golden/harness.*) and (b) the
oracle-backed run of the same host code."""
import json
import os

import numpy as np
import pytest
import torch

import synthetic_data as sdata
from oracle_engine import OracleEngine

from fashionern_aaai2024_amd import synth
from fashionern_aaai2024_amd.clip_model import FernCLIP, create_model
from fashionern_aaai2024_amd.fusion_model import CombinerSimple, DVR_module, VisualSR
from fashionern_aaai2024_amd.model import ERN
from fashionern_aaai2024_amd.run import _common, test_200k, test_cirr, test_fiq, test_shoes, test_val
from fashionern_aaai2024_amd.tokenizer import register_tokenizer
from fashionern_aaai2024_amd.utils import extract_index_features

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(GOLD, "harness.json")))
ARR = np.load(os.path.join(GOLD, "harness.npz"))
register_tokenizer("stub", sdata.stub_tokenizer)
DEV = "cuda:0"


# The two precisions that claim north_star's parity contract (scores within 1e-3, identical ordering): the exact fp32 fma chain and
# FERN_PREC_F32X3 (fp32 data, large plain GEMMs from three bf16 planes per operand).  Every golden / oracle test below runs under
# both at the SAME tolerances: f32x3 is either inside the contract against the reference-generated goldens
# or it is not.
PARITY_PRECISIONS = ["fp32", "f32x3"]


def build(kind, engine=None, device=DEV, precision="fp32"):
    d, n, q = META["d"], META["n"], META["q"]
    clip = sdata.StubCLIP(d).eval().to(device)
    model = ERN(clip, d, device, engine=engine)
    if precision != "fp32":
        model.engine.set_precision(precision)
    model.load_state_dict(synth.fusion_state_dict(d, seed=META["fusion_seed"]))
    gal = sdata.Gallery(n, d, seed=META["gallery_seed"], dup_names=(kind == "200k"))
    rel = sdata.RelativeDataset(gal, q, "fiq" if kind == "val" else kind, seed=META["relative_seed"])
    feats, names, local = extract_index_features(sdata.ClassicDataset(gal), clip, 13, device, d, num_workers=0)
    return clip, model, rel, feats, names, local, d


@pytest.mark.parametrize("kind,fn", [("fiq", test_fiq.compute_fiq_val_metrics), ("cirr", test_cirr.compute_cirr_val_metrics),
                                     ("200k", test_200k.compute_200k_val_metrics), ("shoes", test_shoes.compute_shoes_val_metrics),
                                     ("val", test_val.compute_fiq_val_metrics)])
@pytest.mark.parametrize("precision", PARITY_PRECISIONS)
def test_harness_on_hip_reproduces_reference_recalls(kind, fn, precision):
    """The recall tuples of the imported reference run (tests/golden/harness.json), in both arithmetic modes that claim north_star's
    parity contract: the exact fp32 chain and f32x3."""
    clip, model, rel, feats, names, local, d = build(kind, precision=precision)
    res = fn(rel, clip, feats, local, names, model, DEV, d, META["batch_size"], 0, "stub")
    assert list(res) == META["recalls"][kind], (res, META["recalls"][kind])


@pytest.mark.parametrize("precision", PARITY_PRECISIONS)
def test_query_features_scores_and_top50_match_reference_run(precision):
    clip, model, rel, feats, names, local, d = build("fiq", precision=precision)
    assert model.engine.precision == precision
    pred, _ = test_fiq.generate_fiq_val_predictions(clip, rel, model, names, feats, DEV, d, META["batch_size"], 0, "stub")
    assert np.abs(pred.cpu().numpy() - ARR["fiq_predicted"]).max() < 5e-5
    fused = _common.fuse_index(model, feats, local)
    assert np.abs(fused.cpu().numpy() - ARR["fiq_index_fused"]).max() < 5e-5
    scores, idx = model.engine.sim_topk(pred, fused, 50)
    ref_scores = np.take_along_axis(ARR["fiq_predicted"] @ ARR["fiq_index_fused"].T, ARR["fiq_top50"].astype(np.int64), axis=1)
    assert np.abs(scores.cpu().numpy() - ref_scores).max() < 1e-3          # north_star: cosine scores within 1e-3
    same = idx.cpu().numpy() == ARR["fiq_top50"]
    # identical ordering; a swap is only tolerated between neighbours the reference itself separates by < 1e-5
    for r, c in zip(*np.nonzero(~same)):
        full = ARR["fiq_predicted"][r] @ ARR["fiq_index_fused"].T
        assert abs(full[idx[r, c].item()] - full[ARR["fiq_top50"][r, c]]) < 1e-5


def test_full_pipeline_with_hip_clip_matches_oracle_pipeline():
    """FernCLIP (tiny shape) + ERN + FIQ harness on the GPU vs the same host code on the oracle engine."""
    cfg = synth.CLIP_CONFIGS["tiny"]
    d = cfg.embed_dim
    register_tokenizer("tiny", sdata.stub_tokenizer)
    r = np.random.default_rng(5)
    n, q = 300, 40
    gal = sdata.Gallery(n, d, seed=21, image_size=cfg.image_size)
    rel = sdata.RelativeDataset(gal, q, "fiq", seed=22)
    outs = []
    for engine, device in ((None, DEV), (OracleEngine(), "cpu")):
        clip = create_model(cfg, device=device, seed=9, engine=engine)
        model = ERN(clip, d, device, engine=clip.engine).init_random(4)
        feats, names, local = extract_index_features(sdata.ClassicDataset(gal), clip, 13, device, d, num_workers=0)
        pred, targets = test_fiq.generate_fiq_val_predictions(clip, rel, model, names, feats, device, d, 16, 0, "tiny")
        fused = _common.fuse_index(model, feats, local)
        s, i = model.engine.sim_topk(pred, fused, 50)
        rec = test_fiq.compute_fiq_val_metrics(rel, clip, feats, local, names, model, device, d, 16, 0, "tiny")
        outs.append((feats.cpu(), pred.cpu(), fused.cpu(), s.cpu(), i.cpu(), rec))
    (f0, p0, g0, s0, i0, r0), (f1, p1, g1, s1, i1, r1) = outs
    assert (f0 - f1).abs().max() < 2e-4 * max(1.0, f1.abs().max().item())
    assert (p0 - p1).abs().max() < 1e-4 and (g0 - g1).abs().max() < 1e-4
    assert (s0 - s1).abs().max() < 1e-3 and r0 == r1
    full = p1 @ g1.T
    for row, col in zip(*np.nonzero((i0 != i1).numpy())):
        assert abs(full[row, i0[row, col]].item() - full[row, i1[row, col]].item()) < 1e-4


def test_precision_switch_between_two_harness_calls_reaches_every_lane():
    """the harness caches a multi-lane pipeline on the engine and a forked context copies the parent's precision only when it is
    made.  Switching the engine's precision BETWEEN two `generate_fiq_val_predictions` calls must change all lanes: the second call
    equals a call-by-call run (FERN_HARNESS_LANES=0, the parent context alone) at the new precision, batch by batch."""
    cfg = synth.CLIP_CONFIGS["tiny-hd64"]
    d = cfg.embed_dim
    register_tokenizer("tiny-hd64", lambda texts, context_length=77: sdata.stub_tokenizer(texts, context_length, vocab=cfg.vocab_size))
    gal = sdata.Gallery(200, d, seed=31, image_size=cfg.image_size)
    rel = sdata.RelativeDataset(gal, 128, "fiq", seed=32)          # 8 batches of 16: two per lane
    clip = create_model(cfg, device=DEV, seed=9)
    model = ERN(clip, d, DEV, engine=clip.engine).init_random(4)
    eng = model.engine
    feats, names, _ = extract_index_features(sdata.ClassicDataset(gal), clip, 13, DEV, d, num_workers=0)

    def predictions():
        return test_fiq.generate_fiq_val_predictions(clip, rel, model, names, feats, DEV, d, 16, 0, "tiny-hd64")[0].clone()

    eng.set_precision("fp32")
    p_fp32 = predictions()                                          # builds and caches the pipeline at fp32
    lanes = int(os.environ.get("FERN_HARNESS_LANES", "4"))
    assert len(eng._harness_pipe.engines) == lanes
    eng.set_precision("bf16")                                       # the parent context only
    p_bf16 = predictions()
    assert [e.precision for e in eng._harness_pipe.engines] == ["bf16"] * lanes
    os.environ["FERN_HARNESS_LANES"] = "0"
    try:
        p_bf16_serial = predictions()
    finally:
        os.environ.pop("FERN_HARNESS_LANES")
    eng.set_precision("fp32")
    assert torch.equal(p_bf16, p_bf16_serial), "a lane kept the precision it was forked with"
    for lo in range(0, 128, 16):                                    # every batch (= every lane) moved away from its fp32 bits
        assert not torch.equal(p_bf16[lo:lo + 16], p_fp32[lo:lo + 16])
    assert torch.equal(predictions(), p_fp32)                       # and back


def test_standalone_modules_on_hip():
    d = 128
    gold = np.load(os.path.join(GOLD, "fusion.npz"))
    sub = lambda p: {k[len(p):]: v for k, v in synth.fusion_state_dict(d, seed=11).items() if k.startswith(p)}  # noqa: E731
    raw, loc = torch.from_numpy(synth.global_feats(6, d, 42, "ir")), torch.from_numpy(synth.local_feats(6, d, 42, "il"))
    txt = torch.from_numpy(synth.global_feats(4, d, 42, "rg")).repeat(2, 1)[:6]
    comb = CombinerSimple(d, 4 * d, 8 * d, device=DEV).load_state_dict(sub("Combiner_module."))
    assert np.abs(comb(raw, txt).cpu().numpy() - gold["d128_combiner_target"]).max() < 2e-5
    sr = VisualSR(d, device=DEV).load_state_dict(sub("SR_module."))
    assert np.abs(sr(loc).cpu().numpy() - gold["d128_sr_target"]).max() < 2e-5
    dvr = DVR_module(d, device=DEV).load_state_dict(sub("DVR."))
    rl, ts = torch.from_numpy(synth.local_feats(4, d, 42, "rl")), torch.from_numpy(synth._normal(42, f"tseq/{d}", (4, 77, d)))
    rg, tg = torch.from_numpy(synth.global_feats(4, d, 42, "rg")), torch.from_numpy(synth.global_feats(4, d, 42, "tg"))
    assert np.abs(dvr(rl, ts, rg, tg).cpu().numpy() - gold["d128_dvr_module"]).max() < 5e-5
    from fashionern_aaai2024_amd._lib import FernError
    with pytest.raises(FernError, match="not finalised"):
        comb.engine.visual_sr(0, loc)          # only the Combiner part was finalised on that context


@pytest.mark.parametrize("precision", PARITY_PRECISIONS)
@pytest.mark.parametrize("d", [128, 512, 640])
def test_ern_matches_reference_goldens(d, precision):
    gold = np.load(os.path.join(GOLD, "fusion.npz"))
    model = ERN(None, d, DEV).load_state_dict(synth.fusion_state_dict(d, seed=11))
    model.engine.set_precision(precision)      # mode="test": 4 x 91 = 364 token rows -> the BERT GEMMs run split under f32x3
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    rg, rl = t(synth.global_feats(4, d, 42, "rg")), t(synth.local_feats(4, d, 42, "rl"))
    tg, ts = t(synth.global_feats(4, d, 42, "tg")), t(synth._normal(42, f"tseq/{d}", (4, 77, d)))
    raw, loc = t(synth.global_feats(6, d, 42, "ir")), t(synth.local_feats(6, d, 42, "il"))
    out = model(ref_feats=rg, ref_local_feats=rl, text_feats=tg, text_seq_feats=ts, mode="test")
    assert np.abs(out.cpu().numpy() - gold[f"d{d}_test"]).max() < 5e-5
    idx = model(tar_feats=torch.nn.functional.normalize(raw, dim=-1), tar_local_feats=loc, mode="index")
    assert np.abs(idx.cpu().numpy() - gold[f"d{d}_index"]).max() < 2e-5


@pytest.mark.parametrize("precision", PARITY_PRECISIONS)
def test_clip_matches_in_tree_reference_statement(precision):
    gold = np.load(os.path.join(GOLD, "clip.npz"))
    for name, n_img, n_txt, tol in (("tiny", 5, 6, 2e-4), ("tiny-hd64", 5, 6, 2e-4), ("ViT-B-16", 2, 2, 1e-3)):
        cfg = synth.CLIP_CONFIGS[name]
        clip = create_model(cfg, device=DEV, seed=5, precision=precision)
        assert clip.engine.precision == precision
        img = clip.encode_image(torch.from_numpy(synth.images(n_img, cfg, 42)))
        assert np.abs(img.cpu().numpy() - gold[f"{name}_image"]).max() < tol
        for tag, full in (("full", True), ("ragged", False)):
            toks = torch.from_numpy(synth.captions(n_txt, cfg, 42, full_length=full))
            g, s = clip.encode_text(toks)
            assert np.abs(s.cpu().numpy() - gold[f"{name}_text_{tag}_seq"]).max() < tol
            assert np.abs(g.cpu().numpy() - gold[f"{name}_text_{tag}_global"]).max() < tol
            assert torch.equal(clip.encode_text(toks, mode="seq"), s)
        clip.engine.close()


@pytest.mark.parametrize("cdim", [64, 640])
def test_clip4cir_combiner_and_element_wise_sum(cdim):
    from fashionern_aaai2024_amd.others import Combiner, element_wise_sum
    gold = np.load(os.path.join(GOLD, "fusion.npz"))
    comb = Combiner(cdim, 4 * cdim, 8 * cdim, device=DEV).load_state_dict(synth.clip4cir_state_dict(cdim, 4 * cdim, 8 * cdim, seed=11))
    im, tx = torch.from_numpy(synth.global_feats(5, 2 * cdim, 42, "c4i")), torch.from_numpy(synth.global_feats(5, 2 * cdim, 42, "c4t"))
    assert np.abs(comb(im, tx).cpu().numpy() - gold[f"clip4cir_c{cdim}"]).max() < 2e-5
    assert np.abs(element_wise_sum(im.cuda(), tx.cuda(), engine=comb.engine).cpu().numpy() - gold[f"ews_c{cdim}"]).max() < 1e-6


_HEADLINE_ORACLE = {}


def _headline_oracle():
    """Inputs, weights and the CPU oracle's outputs of the headline-shape test, computed once for both precisions."""
    if _HEADLINE_ORACLE:
        return _HEADLINE_ORACLE
    from oracle import clip as oclip, fusion as ofusion, rank as orank
    cfg = synth.CLIP_CONFIGS["ViT-B-16"]
    d, b, n, k = cfg.embed_dim, 64, 46_000, 50
    clip_sd, fusion_sd = synth.clip_state_dict(cfg, seed=0), synth.fusion_state_dict(d, seed=0)
    im, tk = torch.from_numpy(synth.images(b, cfg, 42)), torch.from_numpy(synth.captions(b, cfg, 42))
    lc = torch.from_numpy(synth.local_feats(b, d, 42))
    graw, gloc = torch.from_numpy(synth.global_feats(n, d, tag="hg")), torch.from_numpy(synth.local_feats(n, d, tag="hgl"))
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    csd, fsd = ofusion.as_torch(clip_sd), ofusion.as_torch(fusion_sd)
    with torch.no_grad():
        orf = oclip.encode_image(csd, cfg, im)
        otg, ots = oclip.encode_text(csd, cfg, tk)
        oq = ofusion.dvr_fuse(fsd, lc, ots, orf, otg)
        ogal = torch.cat([ofusion.index_fuse(fsd, torch.nn.functional.normalize(graw[o:o + 4096], dim=-1), gloc[o:o + 4096]) for o in range(0, n, 4096)])
        full = oq @ ogal.T
        os_, oi = orank.cosine_topk(oq, ogal, k)
    _HEADLINE_ORACLE.update(cfg=cfg, d=d, k=k, clip_sd=clip_sd, fusion_sd=fusion_sd, im=im, tk=tk, lc=lc, graw=graw, gloc=gloc, orf=orf, otg=otg,
                            ots=ots, oq=oq, ogal=ogal, full=full, os=os_, oi=oi)
    return _HEADLINE_ORACLE


@pytest.mark.parametrize("precision", PARITY_PRECISIONS)
def test_headline_shape_b64_vit_b16_matches_the_oracle(precision):
    """The bench's headline shape as a test of its own: ONE batch of 64 composed queries, ViT-B/16 image +
    text towers, fusion, top-50 of a 46k-row fused gallery -- HIP vs the fp32 oracle on the same seeded inputs.  Tolerances:
    features <= 1e-3 of the feature scale (north_star), cosine scores <= 1e-3, top-50 ordering identical except between rows
    the oracle itself separates by < 2e-6 (its BLAS summation order is not the kernel's).  Both parity precisions, same bounds
    (VERDICT r3 item 1a); the oracle side is computed once."""
    o = _headline_oracle()
    cfg, d, k = o["cfg"], o["d"], o["k"]
    clip = create_model(cfg, device=DEV, precision=precision)
    clip.load_state_dict(o["clip_sd"])
    model = ERN(clip, d, DEV, engine=clip.engine).load_state_dict(o["fusion_sd"])
    eng = model.engine
    assert eng.precision == precision
    rf = eng.encode_image(o["im"].to(DEV))
    tg, ts = eng.encode_text(o["tk"].to(DEV))
    q = eng.dvr_fuse(rf, o["lc"].to(DEV), tg, ts)
    gal = eng.index_fuse(o["graw"], o["gloc"], normalize_input=True)
    s, i = eng.sim_topk(q, gal, k)
    eng.sync()
    for name, got, want in (("image", rf, o["orf"]), ("text_global", tg, o["otg"]), ("text_seq", ts, o["ots"])):
        assert (got.cpu() - want).abs().max().item() <= 1e-3 * want.abs().max().item(), name
    assert (q.cpu() - o["oq"]).abs().max().item() < 1e-4 and (gal.cpu() - o["ogal"]).abs().max().item() < 1e-4      # unit-norm fused features
    assert (s.cpu() - o["os"]).abs().max().item() < 1e-3
    gi, oi, full = i.cpu().long(), o["oi"], o["full"]
    for r, c in (gi != oi).nonzero().tolist():
        assert abs(full[r, gi[r, c]].item() - full[r, oi[r, c]].item()) < 2e-6, (r, c)
    eng.close()


def test_cli_driver_runs_on_synthetic_split():
    """python -m fashionern_aaai2024_amd.run.test_fiq: the reference driver's flags on a seeded synthetic split."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mod, extra in (("test_fiq", []), ("test_cirr", [])):
        r = subprocess.run([sys.executable, "-m", f"fashionern_aaai2024_amd.run.{mod}", "--clip-model-name", "tiny", "--feature-dim", "128",
                            "--input-dim", "64", "--synthetic-gallery", "300", "--synthetic-queries", "40", "--batch-size", "16"] + extra,
                           cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Average:" in r.stdout


def test_cli_driver_under_torchrun_prints_the_single_process_recalls():
    """run/test_{fiq,cirr,200k} under torch.distributed.run, 2 ranks (sharing this box's one GPU over gloo -- the debug layout;
    on a multi-GPU node the same command takes cuda:LOCAL_RANK over RCCL): sharded gallery encode + fuse + all-gather, query data
    parallel -- rank 0 prints exactly the lines of the single-process run."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--clip-model-name", "tiny", "--feature-dim", "128", "--input-dim", "64", "--synthetic-gallery", "301", "--synthetic-queries", "41",
             "--batch-size", "16"]
    for mod in ("test_fiq", "test_cirr", "test_200k"):
        one = subprocess.run([sys.executable, "-m", f"fashionern_aaai2024_amd.run.{mod}"] + flags, cwd=root, capture_output=True, text=True, timeout=600)
        assert one.returncode == 0, one.stderr[-2000:]
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, FERN_DIST_BACKEND="gloo", FERN_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), "-m", f"fashionern_aaai2024_amd.run.{mod}"] + flags,
                             cwd=root, capture_output=True, text=True, timeout=900, env=env)
        assert two.returncode == 0, two.stderr[-3000:]
        keep = lambda out: [ln for ln in out.splitlines() if "recalls:" in ln or ln.startswith(("R@", "Average"))]  # noqa: E731
        assert keep(one.stdout) and keep(two.stdout) == keep(one.stdout), (keep(one.stdout), keep(two.stdout))


@pytest.mark.parametrize("cfg_name", ["tiny-w256", "tiny-resnet"])
def test_full_pipeline_in_f32x3_mode_is_fp32_accurate(cfg_name):
    """FERN_PREC_F32X3 end to end (encode image + text, fusion, ranking; ViT and ModifiedResNet image towers -- the ResNet's 1x1
    convolutions run split, its 3x3 window loader stays on the exact kernels): query / gallery features within 2e-5 of the fp32 mode's
    (unit-norm features; measured ~1e-6), cosine scores within 1e-5, and the ranking identical except between rows the fp32 mode itself
    separates by < 2e-6 -- the same near-tie allowance the fp32 mode gets against the BLAS-ordered oracle."""
    cfg = synth.CLIP_CONFIGS[cfg_name]
    d = cfg.embed_dim
    im, tk = torch.from_numpy(synth.images(40, cfg, 7)), torch.from_numpy(synth.captions(40, cfg, 7))
    lc = torch.from_numpy(synth.local_feats(40, d, 7))
    graw, gloc = torch.from_numpy(synth.global_feats(3000, d, tag="xg")), torch.from_numpy(synth.local_feats(3000, d, tag="xgl"))
    out = {}
    for prec in ("fp32", "f32x3"):
        clip = create_model(cfg, device=DEV, seed=9, precision=prec)
        model = ERN(clip, d, DEV, engine=clip.engine).init_random(4)
        eng = model.engine
        assert eng.precision == prec
        rf = eng.encode_image(im.to(DEV))
        tg, ts = eng.encode_text(tk.to(DEV))
        q = eng.dvr_fuse(rf, lc.to(DEV), tg, ts)
        gal = eng.index_fuse(graw, gloc, normalize_input=True)
        s, i = eng.sim_topk(q, gal, 50)
        out[prec] = tuple(t.cpu() for t in (rf, q, gal, s, i))
        eng.close()
    rf0, q0, g0, s0, i0 = out["fp32"]
    rf1, q1, g1, s1, i1 = out["f32x3"]
    assert not torch.equal(rf0, rf1), "the mode was expected to change the arithmetic of the tower GEMMs"
    assert (rf0 - rf1).abs().max().item() <= 2e-5 * rf0.abs().max().item()
    assert (q0 - q1).abs().max().item() < 2e-5 and (g0 - g1).abs().max().item() < 2e-5
    assert (s0 - s1).abs().max().item() < 1e-5
    full = q0.double() @ g0.double().T
    for r, c in (i0 != i1).nonzero().tolist():
        assert abs(full[r, i0[r, c]].item() - full[r, i1[r, c]].item()) < 2e-6, (r, c)


def test_full_pipeline_in_bf16_perf_mode_stays_within_the_score_budget():
    """The whole FIQ harness with the encoders in bf16 perf mode vs the same run in fp32 parity mode.

    The mode moves the CLIP features by ~1e-5 in cosine (an angle of a few 1e-3 rad), so a cosine SCORE against another
    unit vector can move by a few 1e-3: the perf mode is NOT inside north_star's 1e-3 score tolerance -- that contract
    (and exact top-K ordering) belongs to the fp32 mode.  What the perf mode must keep (measured on this split:
    score deviations <= 2.8e-3, top-50 overlap 0.996, identical recall tuples): scores within 5e-3, top-50 overlap
    > 0.95, recall@K within two queries."""
    cfg = synth.CLIP_CONFIGS["tiny-hd64"]
    d = cfg.embed_dim
    register_tokenizer("tiny-hd64", lambda texts, context_length=77: sdata.stub_tokenizer(texts, context_length, vocab=cfg.vocab_size))
    gal = sdata.Gallery(300, d, seed=31, image_size=cfg.image_size)
    rel = sdata.RelativeDataset(gal, 40, "fiq", seed=32)
    outs = []
    for precision in ("fp32", "bf16"):
        clip = create_model(cfg, device=DEV, seed=9, precision=precision)
        assert clip.engine.precision == precision
        model = ERN(clip, d, DEV, engine=clip.engine).init_random(4)
        feats, names, local = extract_index_features(sdata.ClassicDataset(gal), clip, 13, DEV, d, num_workers=0)
        pred, _ = test_fiq.generate_fiq_val_predictions(clip, rel, model, names, feats, DEV, d, 16, 0, "tiny-hd64")
        fused = _common.fuse_index(model, feats, local)
        s, i = model.engine.sim_topk(pred, fused, 50)
        rec = test_fiq.compute_fiq_val_metrics(rel, clip, feats, local, names, model, DEV, d, 16, 0, "tiny-hd64")
        outs.append((pred.cpu(), s.cpu(), i.cpu(), rec))
    (p0, s0, i0, r0), (p1, s1, i1, r1) = outs
    assert not torch.equal(p0, p1)                                      # really another precision
    assert (1 - torch.nn.functional.cosine_similarity(p0, p1, dim=-1)).max() < 1e-4
    assert (s0[:, 0] - s1[:, 0]).abs().max() < 5e-3
    overlap = np.mean([len(set(a.tolist()) & set(b.tolist())) / 50 for a, b in zip(i0, i1)])
    assert overlap > 0.95
    assert all(abs(a - b) <= 5.0 for a, b in zip(r0, r1))               # recall@K in percent, 40 queries: <= 2 queries apart


def test_bf16_precision_is_refused_for_the_resnet_tower():
    with pytest.raises(ValueError, match="transformer towers"):
        create_model("tiny-resnet", device=DEV, seed=1, precision="bf16")


def test_batch_classification_loss_and_train_mode_match_the_reference():
    """losses/loss.py:10-14 and ERN's default ("train") mode (models/model.py:71-75) on the HIP engine, against the values
    the imported reference produced (tests/golden/loss.npz)."""
    from fashionern_aaai2024_amd.losses import BatchBasedClassificationLoss
    z = np.load(os.path.join(GOLD, "loss.npz"))
    d = 128
    clip = sdata.StubCLIP(d).eval().to(DEV)
    model = ERN(clip, d, DEV)
    model.load_state_dict(synth.fusion_state_dict(d, seed=11))
    crit = BatchBasedClassificationLoss(model.engine)
    for key in [k for k in z.files if k.startswith("loss_")]:
        _, b, dd = key.split("_")
        got = crit(torch.from_numpy(z[f"pred_{b}_{dd}"]), torch.from_numpy(z[f"tar_{b}_{dd}"]))
        # logits are 100 x cosines: an fp32 rounding of a logit is ~1e-5, so is the loss
        assert abs(got.item() - float(z[key])) < 2e-4 * max(1.0, abs(float(z[key]))), key
    t = lambda k: torch.from_numpy(z["train_in_" + k])  # noqa: E731
    fusion, target = model(ref_feats=t("ref"), ref_local_feats=t("loc"), text_feats=t("tg"), text_seq_feats=t("ts"),
                           tar_feats=t("tar"), tar_local_feats=t("tloc"))
    assert (fusion.cpu() - torch.from_numpy(z["train_fusion"])).abs().max() < 5e-5
    assert (target.cpu() - torch.from_numpy(z["train_target"])).abs().max() < 5e-5
    assert abs(crit(fusion, target).item() - float(z["train_loss"])) < 5e-3


def test_cli_driver_reads_real_layouts_from_data_root(tmp_path):
    """`python -m ...run.test_fiq/test_cirr/test_shoes --data-root DIR`: the file-backed dataset classes, GPU preprocessing of
    decoded images (TargetPad / bicubic / crop / normalise kernels), the built-in BPE tokenizer (FERN_CLIP_BPE_VOCAB) and the
    HIP engine, end to end on a synthetic directory tree in the reference's layouts."""
    import gzip
    import subprocess
    import sys
    cfg = synth.CLIP_CONFIGS["tiny"]
    root = sdata.write_dataset_tree(tmp_path / "data", cfg.embed_dim)
    vocab = tmp_path / "bpe.txt.gz"
    with gzip.open(vocab, "wt", encoding="utf-8") as f:
        f.write("#version: test\n" + "\n".join(["l o", "lo n", "lon g</w>", "r e", "m o", "mo re</w>"]) + "\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FERN_CLIP_BPE_VOCAB=str(vocab))
    common = ["--clip-model-name", "tiny", "--feature-dim", str(cfg.embed_dim), "--input-dim", str(cfg.image_size), "--batch-size", "4"]
    for mod, data_root, marker in (("test_fiq", root, "R@10:"), ("test_cirr", root, "Average:"), ("test_shoes", os.path.join(root, "shoes"), "R@10:"),
                                   ("test_200k", os.path.join(root, "fashion200k"), "R@10:")):
        r = subprocess.run([sys.executable, "-m", f"fashionern_aaai2024_amd.run.{mod}", "--data-root", data_root] + common,
                           cwd=repo, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert marker in r.stdout and "recalls:" in r.stdout, r.stdout[-500:]


def test_extract_patch_driver_writes_the_features_the_datasets_read(tmp_path):
    """`python -m ...run.extract_patch`: images under a directory -> <name>.pth with [13, D] float32 features; the files are what
    FashionIQDataset loads, equal what `preprocess.extract_patch_features` returns in-process, and a second run skips them."""
    import subprocess
    import sys
    from PIL import Image
    from fashionern_aaai2024_amd.dataloader import FashionIQDataset
    from fashionern_aaai2024_amd.preprocess import extract_patch_features
    cfg = synth.CLIP_CONFIGS["tiny"]
    root = sdata.write_dataset_tree(tmp_path / "data", cfg.embed_dim)
    images = os.path.join(root, "fashion-iq", "images")
    out = os.path.join(root, "fashion-iq", "my_local13")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "fashionern_aaai2024_amd.run.extract_patch", "--images", images, "--out", out, "--clip-model-name", "tiny", "--seed", "7"]
    r = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "36 feature files written" in r.stdout, r.stderr[-2000:] + r.stdout[-500:]
    r2 = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and "0 feature files written" in r2.stdout
    ds = FashionIQDataset("val", ["dress"], "classic", lambda im: torch.zeros(3, 4, 4), base_path=root, local_dir="my_local13", strict=True)
    name, _, loc = ds[5]
    assert loc.dtype == torch.float32 and tuple(loc.shape) == (13, cfg.embed_dim)
    clip = create_model(cfg, device=DEV, seed=7)
    img = torch.from_numpy(np.array(Image.open(os.path.join(images, f"{name}.png")).convert("RGB"), dtype=np.uint8)).to(DEV)
    assert torch.equal(extract_patch_features(clip, img).cpu(), loc)
