"""`MGFNLoss` with the reference's class name and signature (`/root/reference/src/loss/mgfn.py:7-47`), as plain torch ops:
an API-compatibility module for user code that instantiates the loss classes directly.  The product path does NOT go
through it: `MGFNForVideoAnomalyDetection.forward` computes all four loss terms with the fused HIP kernels of
csrc/loss.hip (`mil_ops.mgfn_loss`, forward and backward)."""
import torch
from torch import nn

from .base import ContrastiveLoss


class MGFNLoss(nn.Module):
    def __init__(self, alpha: float = 0.001):
        super().__init__()
        self.alpha = alpha
        self.criterion = nn.BCELoss()
        self.contrastive = ContrastiveLoss()

    def forward(self, abnormal_scores, normal_scores, a_feat_magnitude, n_feat_magnitude, abnormal_labels, normal_labels):
        labels = torch.cat((normal_labels, abnormal_labels), 0)
        scores = torch.cat((normal_scores, abnormal_scores), 0).squeeze()
        sep = int(len(n_feat_magnitude) / 2)
        l1 = lambda v: torch.linalg.vector_norm(v, ord=1, dim=2)
        loss_cls = self.criterion(scores, labels)
        loss_con = self.contrastive(l1(a_feat_magnitude), l1(n_feat_magnitude), 1)
        loss_con_n = self.contrastive(l1(n_feat_magnitude[sep:]), l1(n_feat_magnitude[:sep]), 0)
        loss_con_a = self.contrastive(l1(a_feat_magnitude[sep:]), l1(a_feat_magnitude[:sep]), 0)
        return loss_cls + self.alpha * (self.alpha * loss_con + loss_con_a + loss_con_n)
