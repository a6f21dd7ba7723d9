"""Stand-alone loss modules with the reference's class names and call signatures
(`/root/reference/src/loss/base.py`).  They exist for API compatibility (users who call the
losses directly); `MGFNForVideoAnomalyDetection.forward` does NOT use them -- it computes all four
terms in one fused HIP launch (csrc/loss.hip, `mil_ops.mgfn_loss`)."""
from typing import Optional

import torch
from torch import nn


class TemporalSmoothnessLoss(nn.Module):
    """lambda1 * sum_t (s[t+1] - s[t])^2 over (bs, T, 1) scores (base.py:7-18)."""

    def __init__(self, lambda1: float = 8e-4):
        super().__init__()
        self.lambda1 = lambda1

    def forward(self, x: torch.Tensor, lambda1: Optional[float] = None) -> torch.Tensor:
        lam = self.lambda1 if lambda1 is None else lambda1
        return lam * torch.diff(x, dim=1).square().sum()


class SparsityLoss(nn.Module):
    """lambda2 * mean(||x||_2 over dim 0) (base.py:21-31)."""

    def __init__(self, lambda2: float = 8e-3):
        super().__init__()
        self.lambda2 = lambda2

    def forward(self, x: torch.Tensor, lambda2: Optional[float] = None) -> torch.Tensor:
        lam = self.lambda2 if lambda2 is None else lambda2
        return lam * torch.linalg.vector_norm(x, dim=0).mean()


class ContrastiveLoss(nn.Module):
    """mean((1-y) d^2 + y clamp(margin - d, 0)^2), d = ||o1 - o2 + 1e-6||_2 (base.py:34-48)."""

    def __init__(self, margin: float = 200.0):
        super().__init__()
        self.margin = margin

    def forward(self, output1: torch.Tensor, output2: torch.Tensor, label) -> torch.Tensor:
        d = torch.linalg.vector_norm(output1 - output2 + 1e-6, dim=-1, keepdim=True)
        return ((1 - label) * d.square() + label * (self.margin - d).clamp_min(0.0).square()).mean()
