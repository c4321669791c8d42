"""In-memory synthetic datasets, a stub CLIP and a stub tokenizer for harness-level tests.

They yield exactly the tuple formats the reference's datasets yield (e.g.
/root/reference/dataloader/fashioniq.py:82-100, cirr.py:70-88, shoes.py:40-50,
fashion200k_patch.py:293,354), so the same objects can be fed to the imported reference harness
(tools/make_goldens.py, dev container only) and to this repo's harness.  Everything is derived from
numpy Generators with fixed seeds.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch
from torch.utils.data import Dataset

P = 13
WORDS = ("red", "blue", "longer", "shorter", "sleeves", "striped", "floral", "darker", "brighter", "collar", "with", "is",
         "more", "less", "formal", "casual", "pattern", "plain", "v-neck", "buttons")


def _rng(seed, tag):
    return np.random.default_rng([seed, zlib.crc32(tag.encode())])


def stub_tokenizer(texts, context_length=77, vocab=1000):
    """Deterministic stand-in for open_clip.get_tokenizer(name): (list[str] | str, context_length) -> int64 [B, ctx].
    `vocab` must not exceed the model's vocabulary: out-of-range ids raise (IndexError / FERN_ERR_ARG), as nn.Embedding does."""
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [1 + zlib.crc32(w.encode()) % (vocab - 3) for w in t.lower().split()][: context_length - 2]
        out[i, 0] = vocab - 2
        out[i, 1:1 + len(ids)] = torch.tensor(ids, dtype=torch.long)
        out[i, 1 + len(ids)] = vocab - 1
    return out


class StubCLIP(torch.nn.Module):
    """Object with the call surface of the reference's external clip_model (models/clip_model.py:10-31)."""

    def __init__(self, d, image_size=8, seed=3):
        super().__init__()
        r = _rng(seed, "stubclip")
        self.d = d
        self.wimg = torch.nn.Parameter(torch.from_numpy(r.standard_normal((3 * image_size * image_size, d)).astype(np.float32) * 0.1), False)
        self.emb = torch.nn.Parameter(torch.from_numpy(r.standard_normal((1000, d)).astype(np.float32)), False)
        self.pos = torch.nn.Parameter(torch.from_numpy(r.standard_normal((77, d)).astype(np.float32) * 0.3), False)

    def encode_image(self, images):
        return images.flatten(1).float() @ self.wimg

    def encode_text(self, text, mode="global", visual_emb=None):
        assert visual_emb is None or visual_emb.shape[0] == P
        seq = self.emb[text.to(self.emb.device)] + self.pos
        if mode == "seq":
            return seq
        pooled = seq[torch.arange(text.shape[0]), text.argmax(dim=-1)]
        return pooled, seq


def _caption(r):
    return " ".join(r.choice(WORDS, size=int(r.integers(2, 6))))


class Gallery:
    """Shared pool: names, tiny images and 13 x D local features."""

    def __init__(self, n, d, seed, image_size=8, dup_names=False):
        r = _rng(seed, "gallery")
        self.n, self.d = n, d
        self.images = r.standard_normal((n, 3, image_size, image_size)).astype(np.float32)
        self.local = r.standard_normal((n, P, d)).astype(np.float32)
        if dup_names:       # Fashion200k: gallery "names" are caption ids shared by several rows
            self.names = [f"cap{int(i)}" for i in r.integers(0, max(2, n // 3), size=n)]
        else:
            self.names = [f"img{i:05d}" for i in range(n)]


class ClassicDataset(Dataset):
    def __init__(self, gal: Gallery):
        self.g = gal

    def __len__(self):
        return self.g.n

    def __getitem__(self, i):
        return self.g.names[i], torch.from_numpy(self.g.images[i]), torch.from_numpy(self.g.local[i])


class RelativeDataset(Dataset):
    """kind in {"fiq", "cirr", "shoes", "200k"}; items follow the reference's per-dataset tuple layout."""

    def __init__(self, gal: Gallery, q, kind, seed):
        r = _rng(seed, "relative/" + kind)
        self.g, self.kind, self.items = gal, kind, []
        for _ in range(q):
            ref, tgt = (int(v) for v in r.choice(gal.n, size=2, replace=False))
            patch = torch.from_numpy(gal.local[ref])
            if kind == "fiq":
                caps = [_caption(r) + ".", " " + _caption(r) + "?"]
                self.items.append((gal.names[ref], gal.names[tgt], caps, patch))
            elif kind == "cirr":
                others = [int(v) for v in r.choice([i for i in range(gal.n) if i not in (ref, tgt)], size=4, replace=False)]
                members = [gal.names[ref], gal.names[tgt]] + [gal.names[o] for o in others]
                order = r.permutation(6)
                self.items.append((gal.names[ref], gal.names[tgt], _caption(r), patch, [members[int(o)] for o in order]))
            elif kind == "shoes":
                self.items.append((gal.names[ref], gal.names[tgt], _caption(r), patch, torch.from_numpy(gal.local[tgt])))
            elif kind == "200k":
                self.items.append((torch.from_numpy(gal.images[ref]), gal.names[ref], _caption(r), gal.names[tgt],
                                   int(r.integers(1, 9)), patch))
            else:
                raise ValueError(kind)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


# ---- a synthetic directory tree in the reference's dataset layouts (dataloader/fashioniq.py, cirr.py, shoes.py) -----------
def _write_image(path, seed):
    import os
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 256, size=(40, 30, 3), dtype=np.uint8)).save(path)


def _write_local(path, seed, d):
    import os
    from fashionern_aaai2024_amd import synth
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(torch.from_numpy(synth._normal(seed, "loc", (13, d))), path)


def write_dataset_tree(root, d, local_dir="fashion_local13"):
    """fashion-iq/, cirr_dataset/ and shoes/ under `root` with 12 / 14 / 10 images, their [13, d] local features and the JSON files."""
    import json
    import os
    root = str(root)
    fiq = os.path.join(root, "fashion-iq")
    for sub in ("captions", "image_splits"):
        os.makedirs(os.path.join(fiq, sub), exist_ok=True)
    for ti, t in enumerate(("dress", "shirt", "toptee")):
        names = [f"{t}{i:03d}" for i in range(12)]
        for i, n in enumerate(names):
            _write_image(os.path.join(fiq, "images", f"{n}.png"), 100 * ti + i)
            _write_local(os.path.join(fiq, local_dir, f"{n}.pth"), 100 * ti + i, d)
        trips = [{"candidate": names[i], "target": names[(i * 5 + 3) % 12], "captions": [f"is more {t} like.", "has longer sleeves?"]}
                 for i in range(8)]
        for split in ("val", "train", "test"):
            json.dump(trips, open(os.path.join(fiq, "captions", f"cap.{t}.{split}.json"), "w"))
            json.dump(names, open(os.path.join(fiq, "image_splits", f"split.{t}.{split}.json"), "w"))
    cirr = os.path.join(root, "cirr_dataset")
    os.makedirs(os.path.join(cirr, "cirr", "captions"), exist_ok=True)
    os.makedirs(os.path.join(cirr, "cirr", "image_splits"), exist_ok=True)
    cnames = [f"dev-{i}-img" for i in range(14)]
    for i, n in enumerate(cnames):
        _write_image(os.path.join(cirr, "dev", f"{n}.png"), 500 + i)
        _write_local(os.path.join(cirr, "cirr_local_13", f"{n}.pth"), 500 + i, d)
    ctrips = []
    for i in range(9):
        ref, tgt = cnames[i], cnames[(i + 4) % 14]
        members = [ref] + [cnames[(i + 4 + k) % 14] for k in range(5)]      # 6 members: the reference and five others incl. the target
        ctrips.append({"pairid": i, "reference": ref, "target_hard": tgt, "caption": f"make it number {i}", "img_set": {"members": members}})
    for split in ("val", "test1", "train"):
        json.dump(ctrips, open(os.path.join(cirr, "cirr", "captions", f"cap.rc2.{split}.json"), "w"))
        json.dump({n: f"./dev/{n}.png" for n in cnames}, open(os.path.join(cirr, "cirr", "image_splits", f"split.rc2.{split}.json"), "w"))
    shoes = os.path.join(root, "shoes")
    os.makedirs(shoes, exist_ok=True)
    rels = [f"womens_athletic_shoes/{i}/img_womens_athletic_shoes_{i}.jpg" for i in range(10)]
    for i, r in enumerate(rels):
        _write_image(os.path.join(shoes, r), 900 + i)
        _write_local(os.path.join(shoes, "shoes_local_feature_13", r.split("/")[-1].split(".jpg")[0] + ".pth"), 900 + i, d)
    ann = [{"ImageName": rels[(i + 3) % 10], "ReferenceImageName": rels[i], "RelativeCaption": f"are less shiny {i}"} for i in range(7)]
    for split in ("test", "train"):
        json.dump(rels, open(os.path.join(shoes, f"split.{split}.json"), "w"))
        json.dump(ann, open(os.path.join(shoes, f"triplet.{split}.json"), "w"))
    # Fashion200k: <f2k>/{labels/dress_test_detect_all.txt, test_queries.txt, women/.../x.jpeg, <local_dir>/.../x.jpeg.pth}
    f2k = os.path.join(root, "fashion200k")
    os.makedirs(os.path.join(f2k, "labels"), exist_ok=True)
    colours = ["red", "blue", "green", "black"]
    kinds = ["mini dress", "maxi dress & belt", "shirt dress."]
    rows = []
    for i in range(12):
        rel = f"women/dresses/casual/{i}/{i}_0.jpeg"
        cap = f"{colours[i % 4]} {kinds[i % 3]}"
        rows.append((rel, cap))
        _write_image(os.path.join(f2k, rel), 1300 + i)
        for ld in ("local_features", "fashion200k_13_patch"):
            _write_local(os.path.join(f2k, rel.replace("women", ld) + ".pth"), 1300 + i, d)
    with open(os.path.join(f2k, "labels", "dress_test_detect_all.txt"), "w", encoding="utf8") as f:
        for rel, cap in rows:
            f.write(f"{rel}\t0.9\t{cap}\n")
    with open(os.path.join(f2k, "test_queries.txt"), "w") as f:
        for i in range(8):
            f.write(f"{rows[i][0]} {rows[(i + 1) % 12][0]}\n")
    return root
