"""ERN: the reference's model wrapper (/root/reference/models/model.py:7-75) on the HIP path.

Same constructor (``ERN(clip_model, feature_dim, device)``), same six ``mode=`` values, same state-dict key
names.  ``mode="train"`` returns the same pair of features as the reference's forward in
eval mode (dropout / BatchNorm batch statistics, i.e. actual training, are outside this path).
"""
from __future__ import annotations

from typing import Mapping

import numpy as np
import torch

from . import synth
from .clip_model import ImageCLIP, TextCLIP
from .engine import PART_ALL, FernEngine

_CLIP_PREFIXES = ("image_clip.clip_model.", "text_clip.clip_model.")


class ERN:
    def __init__(self, clip_model, feature_dim, device=None, engine=None):
        self.image_clip = ImageCLIP(clip_model)          # model.py:12
        self.text_clip = TextCLIP(clip_model)            # model.py:13
        self.feature_dim = int(feature_dim)
        self.engine = engine if engine is not None else FernEngine(device if device is not None else "cuda:0")
        self.device = self.engine.device
        self._state = {}

    # -- nn.Module-like surface (test_fiq.py:148-149,168-169) ----------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, object], strict: bool = True):
        """Accepts a full ``ERN.state_dict()``: fusion keys are consumed; ``image_clip.*`` / ``text_clip.*`` are
        forwarded to the wrapped clip_model when it can take them; ``position_ids`` / ``num_batches_tracked`` /
        pooler weights are accepted and unused.

        ``strict`` follows ``nn.Module.load_state_dict`` (test_fiq.py:149 calls it with the default, True): with
        ``strict=True`` a missing or an unexpected fusion key raises ``RuntimeError`` naming the keys; with ``strict=False``
        unexpected keys are dropped and missing ones keep the values of the previous load (an error if there was none).
        Always returns ``self`` (so ``ERN(...).load_state_dict(sd)`` chains in both modes); the two key lists of the last load
        are left in ``self.missing_keys`` / ``self.unexpected_keys`` (what nn.Module returns as a named tuple).  Two keys are optional in both modes
        because real checkpoints differ in them: ``DVR.transformer_layer.cls_token`` (absent from GPU-trained
        checkpoints -> zeros) and ``...embeddings.position_ids`` (a buffer in transformers 4.30.2, gone in later versions)."""
        fusion, clip = {}, {}
        for k, v in state_dict.items():
            arr = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if k.startswith(_CLIP_PREFIXES[0]):
                clip[k[len(_CLIP_PREFIXES[0]):]] = arr
            elif k.startswith(_CLIP_PREFIXES[1]):
                clip.setdefault(k[len(_CLIP_PREFIXES[1]):], arr)
            else:
                fusion[k] = arr
        expected = set(synth.fusion_state_shapes(self.feature_dim))
        optional = {"DVR.transformer_layer.cls_token", "DVR.transformer_layer.bert_encoder.bert_model.embeddings.position_ids"}
        optional |= {k for k in fusion if k.endswith("num_batches_tracked")}
        missing = sorted(k for k in expected - optional if k not in fusion)
        unexpected = sorted(k for k in fusion if k not in expected and k not in optional)
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for ERN: missing key(s): " + ", ".join(missing or ["-"]) +
                               "; unexpected key(s): " + ", ".join(unexpected or ["-"]))
        for k in unexpected:
            del fusion[k]
        still = [k for k in missing if k not in self._state]
        if still:
            raise RuntimeError("load_state_dict(strict=False): no earlier value to keep for missing key(s): " + ", ".join(still))
        merged = dict(self._state) if missing else {}
        merged.update(fusion)
        self.engine.load_tensors(merged)
        self.engine.finalize_fusion(self.feature_dim, PART_ALL)
        self._state = merged
        cm = self.image_clip.clip_model
        if clip and hasattr(cm, "load_state_dict") and getattr(cm, "engine", None) is not None:
            cm.load_state_dict(clip)
        self.missing_keys, self.unexpected_keys = missing, unexpected
        return self

    def init_random(self, seed: int = 0):
        return self.load_state_dict(synth.fusion_state_dict(self.feature_dim, seed))

    def state_dict(self):
        return {k: torch.from_numpy(np.array(v)) for k, v in self._state.items()}

    def eval(self):
        return self

    def float(self):
        return self

    def to(self, *a, **k):
        return self

    # -- forward -----------------------------------------------------------------------------------
    def forward(self, image=None, text=None, ref_feats=None, ref_local_feats=None, text_feats=None, text_seq_feats=None,
                tar_feats=None, tar_local_feats=None, mode="train"):
        if mode == "image":
            return self.image_clip(image)                                                    # model.py:55-56
        if mode == "text_global":
            return self.text_clip(text, mode="global", visual_emb=ref_local_feats)[0]        # :58-59
        if mode == "text_seq":
            return self.text_clip(text, mode="seq", visual_emb=ref_local_feats)              # :61-62
        if mode == "index":
            return self.engine.index_fuse(tar_feats, tar_local_feats)                        # :64-66
        if mode == "test":
            return self.engine.dvr_fuse(ref_feats, ref_local_feats, text_feats, text_seq_feats)   # :68-69
        fusion = self.engine.dvr_fuse(ref_feats, ref_local_feats, text_feats, text_seq_feats)     # :71-75
        return fusion, self.engine.index_fuse(tar_feats, tar_local_feats)

    def __call__(self, *a, **k):
        with torch.no_grad():
            return self.forward(*a, **k)
