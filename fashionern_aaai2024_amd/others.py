"""Adjacent call surface of the path: the CLIP4Cir ``Combiner`` of
/root/reference/models/others/Combiner_Model.py:6-70 and ``utils.element_wise_sum`` (/root/reference/utils/utils.py:133-140),
both as variants of the fused Combiner kernels (GEMM epilogues + gate/normalise tail)."""
from __future__ import annotations

import numpy as np
import torch

from .engine import FernEngine

_default_engine = None


def default_engine(device="cuda:0") -> FernEngine:
    global _default_engine
    if _default_engine is None:
        _default_engine = FernEngine(device)
    return _default_engine


class Combiner:
    """Same constructor / call as the reference class; note its Linear layers take inputs of width 2 * clip_feature_dim
    (Combiner_Model.py:17-18), so image_features / text_features are [n, 2 * clip_feature_dim]."""

    def __init__(self, clip_feature_dim: int, projection_dim: int, hidden_dim: int, device=None, engine=None):
        self.clip_feature_dim, self.projection_dim, self.hidden_dim = int(clip_feature_dim), int(projection_dim), int(hidden_dim)
        self.engine = engine if engine is not None else FernEngine(device or "cuda:0")
        self._state = {}

    def load_state_dict(self, state_dict, strict=True):
        sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in state_dict.items()}
        self.engine.load_tensors(sd, prefix="clip4cir.")
        self.engine.finalize_clip4cir()
        self._state = sd
        return self

    def state_dict(self):
        return {k: torch.from_numpy(np.array(v)) for k, v in self._state.items()}

    def eval(self):
        return self

    def float(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, image_features: torch.Tensor, text_features: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self.engine.combiner_clip4cir(image_features, text_features)

    forward = __call__


def element_wise_sum(image_features: torch.Tensor, text_features: torch.Tensor, engine=None) -> torch.Tensor:
    """Normalised element-wise sum (utils/utils.py:133-140)."""
    eng = engine if engine is not None else default_engine(image_features.device if image_features.is_cuda else "cuda:0")
    return eng.element_wise_sum(image_features, text_features)
