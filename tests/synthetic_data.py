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


def stub_tokenizer(texts, context_length=77):
    """Deterministic stand-in for open_clip.get_tokenizer(name): (list[str] | str, context_length) -> int64 [B, ctx]."""
    if isinstance(texts, str):
        texts = [texts]
    vocab = 1000
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
