"""CombinerSimple / VisualSR / DVR_module behind the reference's constructor and call signatures
(/root/reference/models/fusion_model.py:8-154), executed by libfern's fused HIP kernels.

Each stand-alone module owns a native context and loads its weights under the ERN slot it corresponds to
(``Combiner_module.``, ``SR_module.``, ``DVR.``); state-dict keys are the reference's un-prefixed ones.
"""
from __future__ import annotations

from typing import Mapping

import numpy as np
import torch

from . import synth
from .engine import (COMBINER_TARGET, PART_DVR, PART_TARGET_COMBINER, PART_TARGET_SR, PATCH_NUM, SR_TARGET, FernEngine)


def _np_state(state_dict: Mapping[str, object]):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in state_dict.items()}


class _FernModule:
    _prefix = ""
    _part = 0

    def __init__(self, feature_dim, device=None, engine=None):
        self.feature_dim = int(feature_dim)
        self.engine = engine if engine is not None else FernEngine(device or "cuda:0")
        self.device = self.engine.device
        self._state = {}

    # keys real checkpoints may or may not carry (model.py: ERN.load_state_dict documents both)
    _optional = ("transformer_layer.cls_token", "transformer_layer.bert_encoder.bert_model.embeddings.position_ids")

    def load_state_dict(self, state_dict, strict=True):
        """``strict`` follows ``nn.Module.load_state_dict`` (the reference loads with the default, run/test/test_fiq.py:149), the
        same way ``ERN.load_state_dict`` does: strict -> a missing or unexpected key raises ``RuntimeError`` naming the keys before
        anything reaches the native side; ``strict=False`` drops unexpected keys and keeps the previous load's value of a missing
        one (an error if there was none).  Keys are the module's own, un-prefixed ones.  The two key lists of the last load stay in
        ``self.missing_keys`` / ``self.unexpected_keys``."""
        sd = _np_state(state_dict)
        pre = self._prefix
        expected = {k[len(pre):] for k in synth.fusion_state_shapes(self.feature_dim) if k.startswith(pre)}
        optional = set(self._optional) | {k for k in sd if k.endswith("num_batches_tracked")}
        missing = sorted(k for k in expected - optional if k not in sd)
        unexpected = sorted(k for k in sd if k not in expected and k not in optional)
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing key(s): " + ", ".join(missing or ["-"]) +
                               "; unexpected key(s): " + ", ".join(unexpected or ["-"]))
        for k in unexpected:
            del sd[k]
        still = [k for k in missing if k not in self._state]
        if still:
            raise RuntimeError("load_state_dict(strict=False): no earlier value to keep for missing key(s): " + ", ".join(still))
        merged = dict(self._state) if missing else {}
        merged.update(sd)
        self.engine.load_tensors(merged, prefix=pre)
        self.engine.finalize_fusion(self.feature_dim, self._part)
        self._state = merged
        self.missing_keys, self.unexpected_keys = missing, unexpected
        return self

    def state_dict(self):
        return {k: torch.from_numpy(np.array(v)) for k, v in self._state.items()}

    def eval(self):
        return self

    def float(self):
        return self

    def to(self, *a, **k):
        return self

    def forward(self, *a, **k):
        raise NotImplementedError

    def __call__(self, *a, **k):
        with torch.no_grad():
            return self.forward(*a, **k)


class CombinerSimple(_FernModule):
    """fusion_model.py:58-94: out = normalize(s * text + (1 - s) * image), s = sigmoid(MLP(cat(proj_t, proj_i)))."""
    _prefix, _part = "Combiner_module.", PART_TARGET_COMBINER

    def __init__(self, clip_feature_dim=512, projection_dim=512 * 4, hidden_dim=512 * 8, device=None, engine=None):
        super().__init__(clip_feature_dim, device, engine)
        self.projection_dim, self.hidden_dim = int(projection_dim), int(hidden_dim)

    def forward(self, image_features: torch.Tensor, text_features: torch.Tensor):
        return self.engine.combiner(COMBINER_TARGET, image_features, text_features)


class VisualSR(_FernModule):
    """fusion_model.py:97-154: attention pooling of the 13 patch embeddings."""
    _prefix, _part = "SR_module.", PART_TARGET_SR

    def __init__(self, embed_dim=512, dropout_rate=0.5, num_region=13, device=None, engine=None):
        if num_region != PATCH_NUM:
            raise ValueError("VisualSR is built for 13 regions (BatchNorm1d(13), fusion_model.py:108-110)")
        super().__init__(embed_dim, device, engine)

    def forward(self, local_feature: torch.Tensor):
        return self.engine.visual_sr(SR_TARGET, local_feature)


class DVR_module(_FernModule):
    """fusion_model.py:8-55: joint BERT-style encoder + cross attention (MR) + SR pooling + three combiners."""
    _prefix, _part = "DVR.", PART_DVR

    def __init__(self, feature_dim=640, device=None, engine=None):
        super().__init__(feature_dim, device, engine)

    def forward(self, ref_patch_features, text_seq_features, ref_global_feats, text_global_feats):
        return self.engine.dvr_fuse(ref_global_feats, ref_patch_features, text_global_feats, text_seq_features)
