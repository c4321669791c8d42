"""CPU: the file-backed dataset classes (fashionern_aaai2024_amd/dataloader.py) on a synthetic directory tree written in the
reference's layouts (dataloader/fashioniq.py, cirr.py, shoes.py): tuple formats per (mode, split), unreadable items, and the
FashionIQ / CIRR / Shoes harnesses running end to end from files (test-only OracleEngine as the compute engine)."""
import os
import warnings

import numpy as np
import pytest
import torch

import synthetic_data as sdata
from oracle_engine import OracleEngine

from fashionern_aaai2024_amd import synth
from fashionern_aaai2024_amd.clip_model import create_model
from fashionern_aaai2024_amd.dataloader import CIRRDataset, FashionIQDataset, ShoesDataset
from fashionern_aaai2024_amd.model import ERN
from fashionern_aaai2024_amd.run import test_cirr, test_fiq, test_shoes
from fashionern_aaai2024_amd.tokenizer import register_tokenizer
from fashionern_aaai2024_amd.utils import collate_fn, extract_index_features

CFG = synth.CLIP_CONFIGS["tiny"]
D = CFG.embed_dim


def cpu_preprocess(image):
    """PIL -> [3, S, S] float in the post-Normalize domain (a plain resize is enough here: the PIL-exact pipeline has its own tests)."""
    arr = np.asarray(image.convert("RGB").resize((CFG.image_size, CFG.image_size)), dtype=np.float32) / 255.0
    return torch.from_numpy((arr - 0.45) / 0.27).permute(2, 0, 1).contiguous()


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    return sdata.write_dataset_tree(tmp_path_factory.mktemp("data"), D)


def test_fashioniq_tuple_formats_and_validation(tree):
    classic = FashionIQDataset("val", ["dress", "toptee"], "classic", cpu_preprocess, base_path=tree)
    assert len(classic) == 24
    name, img, loc = classic[13]
    assert name == "toptee001" and tuple(img.shape) == (3, CFG.image_size, CFG.image_size) and tuple(loc.shape) == (13, D)
    rel = FashionIQDataset("val", ["dress"], "relative", cpu_preprocess, base_path=tree)
    ref, tgt, caps, ref_loc = rel[2]
    assert (ref, tgt) == ("dress002", "dress001") and caps == ["is more dress like.", "has longer sleeves?"] and tuple(ref_loc.shape) == (13, D)
    tr = FashionIQDataset("train", ["shirt"], "relative", cpu_preprocess, base_path=tree)[0]
    assert len(tr) == 5 and tr[0].shape == tr[1].shape == (3, CFG.image_size, CFG.image_size) and tr[3].shape == tr[4].shape == (13, D)
    te = FashionIQDataset("test", ["shirt"], "relative", cpu_preprocess, base_path=tree)[1]
    assert te[0] == "shirt001" and tuple(te[1].shape) == (3, CFG.image_size, CFG.image_size) and len(te[2]) == 2
    for bad in (dict(split="dev", dress_types=["dress"]), dict(split="val", dress_types=["hat"]), dict(split="val", dress_types=["dress"], mode="x")):
        with pytest.raises(ValueError):
            FashionIQDataset(**{"mode": "relative", **bad}, preprocess=cpu_preprocess, base_path=tree)


def test_unreadable_items_are_dropped_like_the_reference(tree):
    ds = FashionIQDataset("val", ["dress"], "classic", cpu_preprocess, base_path=tree)
    os.rename(os.path.join(tree, "fashion-iq", "fashion_local13", "dress003.pth"), os.path.join(tree, "fashion-iq", "fashion_local13", "dress003.bak"))
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert ds[3] is None and any("unreadable" in str(x.message) for x in w)
        names, imgs, locs = collate_fn([ds[2], ds[3], ds[4]] if False else [ds[2], None, ds[4]])
        assert list(names) == ["dress002", "dress004"] and imgs.shape[0] == 2
        with pytest.raises(Exception):
            FashionIQDataset("val", ["dress"], "classic", cpu_preprocess, base_path=tree, strict=True)[3]
    finally:
        os.rename(os.path.join(tree, "fashion-iq", "fashion_local13", "dress003.bak"), os.path.join(tree, "fashion-iq", "fashion_local13", "dress003.pth"))


def test_cirr_and_shoes_tuple_formats(tree):
    c = CIRRDataset("val", "classic", cpu_preprocess, base_path=tree)
    assert len(c) == 14 and c[5][0] == "dev-5-img" and tuple(c[5][2].shape) == (13, D)
    ref, tgt, cap, loc, members = CIRRDataset("val", "relative", cpu_preprocess, base_path=tree)[3]
    assert (ref, tgt, cap) == ("dev-3-img", "dev-7-img", "make it number 3") and len(members) == 6 and tgt in members and tuple(loc.shape) == (13, D)
    pid, ref, cap, members = CIRRDataset("test1", "relative", cpu_preprocess, base_path=tree)[4]
    assert pid == 4 and ref == "dev-4-img" and len(members) == 6
    assert len(CIRRDataset("train", "relative", cpu_preprocess, base_path=tree)[0]) == 5
    with pytest.raises(ValueError):
        CIRRDataset("test", "relative", cpu_preprocess, base_path=tree)
    sp = os.path.join(tree, "shoes") + "/"
    s = ShoesDataset("test", "classic", cpu_preprocess, shoes_path=sp)
    assert len(s) == 10 and s[2][0] == "img_womens_athletic_shoes_2" and tuple(s[2][2].shape) == (13, D)
    ref, tgt, cap, rl, tl = ShoesDataset("test", "relative", cpu_preprocess, shoes_path=sp)[1]
    assert (ref, tgt, cap) == ("img_womens_athletic_shoes_1", "img_womens_athletic_shoes_4", "are less shiny 1") and rl.shape == tl.shape == (13, D)
    assert len(ShoesDataset("train", "relative", cpu_preprocess, shoes_path=sp)[0]) == 5


def test_harnesses_run_from_files(tree):
    """extract_index_features + compute_{fiq,cirr,shoes}_val_metrics on the file-backed datasets (tiny CLIP, oracle engine)."""
    register_tokenizer("tiny", sdata.stub_tokenizer)
    clip = create_model(CFG, device="cpu", seed=3, engine=OracleEngine())
    model = ERN(clip, D, "cpu", engine=clip.engine).init_random(5)
    classic = FashionIQDataset("val", ["dress"], "classic", cpu_preprocess, base_path=tree)
    rel = FashionIQDataset("val", ["dress"], "relative", cpu_preprocess, base_path=tree)
    feats, names, local = extract_index_features(classic, clip, 13, "cpu", D, num_workers=0)
    assert feats.shape == (12, D) and local.shape == (12, 13, D) and names == classic.image_names
    r10, r50 = test_fiq.compute_fiq_val_metrics(rel, clip, feats, local, names, model, "cpu", D, 4, 0, "tiny")
    assert 0.0 <= r10 <= r50 <= 100.0 and r50 == 100.0                      # 12-image gallery: every target is inside the top 50
    cc = CIRRDataset("val", "classic", cpu_preprocess, base_path=tree)
    cr = CIRRDataset("val", "relative", cpu_preprocess, base_path=tree)
    feats, names, local = extract_index_features(cc, clip, 13, "cpu", D, num_workers=0)
    out = test_cirr.compute_cirr_val_metrics(cr, clip, feats, local, names, model, "cpu", D, 4, 0, "tiny")
    assert len(out) == 7 and all(0.0 <= v <= 100.0 for v in out) and out[2] >= out[1] >= out[0]      # subset recalls are monotone in K
    sp = os.path.join(tree, "shoes") + "/"
    sc, sr = ShoesDataset("test", "classic", cpu_preprocess, shoes_path=sp), ShoesDataset("test", "relative", cpu_preprocess, shoes_path=sp)
    feats, names, local = extract_index_features(sc, clip, 13, "cpu", D, num_workers=0)
    out = test_shoes.compute_shoes_val_metrics(sr, clip, feats, local, names, model, "cpu", D, 4, 0, "tiny")
    assert all(0.0 <= v <= 100.0 for v in out)


def test_fashion200k_evaluation_datasets_and_harness(tree):
    from fashionern_aaai2024_amd.dataloader import (Fashion200kTestDataset, Fashion200kTestQueryDataset, caption_post_process,
                                                    get_different_word)
    from fashionern_aaai2024_amd.run import test_200k
    assert caption_post_process(" a.b?c&d*e ") == "adotmarkbquestionmarkcandmarkdstarmarke"
    assert get_different_word("red mini dress", "blue mini dress") == ("red", "blue", "replace red with blue")
    assert get_different_word("red dress", "red dress")[2] == "replace dress with dress"          # no differing word: the last words
    root = os.path.join(tree, "fashion200k")
    gal = Fashion200kTestDataset(root, "val", cpu_preprocess)
    qs = Fashion200kTestQueryDataset(root, "val", cpu_preprocess)
    assert len(gal) == 12 and len(qs) == 8
    img_id, img, loc = gal[1]
    assert img_id == "blue maxi dress andmark belt" and tuple(img.shape) == (3, CFG.image_size, CFG.image_size) and tuple(loc.shape) == (13, D)
    ref_img, ref_id, mod, tgt_id, n, ref_loc = qs[0]
    assert (ref_id, tgt_id) == ("red mini dress", "blue maxi dress andmark belt") and mod == "replace red with blue" and n == len(mod)
    assert tuple(ref_img.shape) == (3, CFG.image_size, CFG.image_size) and tuple(ref_loc.shape) == (13, D)
    register_tokenizer("tiny", sdata.stub_tokenizer)
    clip = create_model(CFG, device="cpu", seed=3, engine=OracleEngine())
    model = ERN(clip, D, "cpu", engine=clip.engine).init_random(5)
    feats, names, local = extract_index_features(gal, clip, 13, "cpu", D, num_workers=0)
    assert len(names) == 12 and len(set(names)) == 12
    out = test_200k.compute_200k_val_metrics(qs, clip, feats, local, names, model, "cpu", D, 4, 0, "tiny")
    assert all(0.0 <= v <= 100.0 for v in out)
