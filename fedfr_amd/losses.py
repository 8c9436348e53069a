"""Margin heads with the reference's signatures (reference losses.py:4-45), computed by HIP kernels.

``CosFace(s, m)(cosine, label)`` / ``ArcFace(s, m)(cosine, label)`` return scaled logits; rows whose label
is -1 get no margin (PartialFC convention).  Unlike the reference they do not mutate ``cosine`` in place
(the reference allows but does not require callers to rely on that)."""
from __future__ import annotations

import torch
from torch import nn

from . import _C, ops


class CosFace(nn.Module):
    def __init__(self, s=64.0, m=0.40):
        super().__init__()
        self.s, self.m = float(s), float(m)

    def forward(self, cosine, label):
        return ops.MarginFn.apply(cosine, label, self.s, self.m, False)


class ArcFace(nn.Module):
    """theta = acos(cos) (unclamped, as the reference), theta += m at the target, cos(theta) * s."""

    def __init__(self, s=64.0, m=0.5):
        super().__init__()
        self.s, self.m = float(s), float(m)

    def forward(self, cosine: torch.Tensor, label):
        return ops.MarginFn.apply(cosine, label, self.s, self.m, True)


class BCE_loss(nn.Module):
    """reference losses.py:4-15 (reduction 'sum_mean'): pos (lam/r) log(1+e^-z+1e-8), neg ((1-lam)/r) log(1+e^z+1e-8)."""

    def __init__(self, r=30, lambda_=0.7, reduction="sum_mean"):
        super().__init__()
        self.r, self.lambda_, self.reduction = float(r), float(lambda_), reduction

    def forward(self, logits, gts):
        from .client import bce_loss_from_logits
        return bce_loss_from_logits(logits, gts, self.r, self.lambda_)
