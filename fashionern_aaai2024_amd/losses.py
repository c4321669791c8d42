"""``losses.loss.BatchBasedClassificationLoss`` (/root/reference/losses/loss.py:6-14) -- forward value on the HIP engine.

The reference uses it in its (unreleased) training loop on the pair ERN's default mode returns
(/root/reference/models/model.py:71-75).  Training itself is outside this path ("training-side
forward reuses the same kernels"), so this class computes the loss VALUE only; there is no backward.
"""
from __future__ import annotations

import torch

from .engine import FernEngine
from .others import default_engine


class BatchBasedClassificationLoss:
    def __init__(self, engine: FernEngine = None):
        self.engine = engine

    def forward(self, predicted_features: torch.Tensor, tar_features: torch.Tensor) -> torch.Tensor:
        eng = self.engine if self.engine is not None else default_engine(
            predicted_features.device if predicted_features.is_cuda else "cuda:0")
        return eng.batch_classification_loss(predicted_features, tar_features)

    __call__ = forward
