"""`pcseg.loss.Losses` for the hot path (reference pcseg/loss/__init__.py:15-137).

TASeg's configs use CE (label smoothing, ignore 0) + Lovasz-softmax
(minkunet.py:344-348); those two are implemented.  The optional losses of the reference
(Dice, ELL, WCE, Focal, EQLv2, GroupSoftmax) are dense torch code outside the path and raise.
"""
import torch
import torch.nn as nn
from torch.nn import CrossEntropyLoss

from .lovasz import lovasz_softmax

__all__ = ["Losses", "lovasz_softmax"]

_SUPPORTED = ("CELoss", "LovLoss")
_OUT_OF_SCOPE = ("WCELoss", "ELLLoss", "DiceLossV0", "DiceLossV1", "FocalLoss", "EQLv2", "GroupSoftmax",
                 "GroupSoftmax_fgbg_2")


def cross_entropy_smoothed(logits, target, ignore_index=0, label_smoothing=0.0):
    """nn.CrossEntropyLoss(ignore_index, label_smoothing) for [N, C] logits, reduction 'mean'
    (torch/nn/functional.py cross_entropy: (1 - eps) * mean nll + eps * mean(-sum_c logp / C), both means over
    the rows whose label is not ignored), written with row-parallel tensor ops: torch's 2-d nll kernels reduce
    with a single workgroup (200 us forward + 150 us backward at N = 178k)."""
    logp = torch.log_softmax(logits, dim=1)
    valid = target != ignore_index
    w = valid.to(logp.dtype)
    picked = logp.gather(1, torch.where(valid, target, torch.zeros_like(target)).unsqueeze(1)).squeeze(1)
    n = w.sum()
    loss = -(picked * w).sum() / n
    if label_smoothing > 0.0:
        smooth = -(logp.sum(dim=1) * w).sum() / (n * logits.shape[1])
        loss = (1.0 - label_smoothing) * loss + label_smoothing * smooth
    return loss


class Losses(nn.Module):
    def __init__(self, loss_types: list, loss_weights: list, cls_num_pts: list = None, ignore_index: int = 0,
                 knn: int = 10, label_smoothing: float = 0.0, class_weight=None, class_names=None):
        super().__init__()
        for name in loss_types:
            if name in _OUT_OF_SCOPE:
                raise NotImplementedError(f"loss '{name}' is outside the TASeg hot path built here")
            if name not in _SUPPORTED:
                raise KeyError(name)
        self.loss_types = loss_types
        self.loss_weights = loss_weights
        self.ignore_index = ignore_index
        self.ce_loss = CrossEntropyLoss(ignore_index=ignore_index, weight=class_weight,
                                        label_smoothing=label_smoothing)
        self.label_smoothing, self.class_weight = label_smoothing, class_weight
        self.lov_loss = lovasz_softmax

    def forward(self, input, target, xyz=None, offset=None):
        total = 0
        if "CELoss" in self.loss_types:
            if self.class_weight is None and input.dim() == 2:
                ce = cross_entropy_smoothed(input, target, self.ignore_index, self.label_smoothing)
            else:
                ce = self.ce_loss(input, target)
            total = total + ce * self.loss_weights[self.loss_types.index("CELoss")]
        if "LovLoss" in self.loss_types:
            lov = self.lov_loss(input.softmax(dim=1), target, ignore=self.ignore_index)
            total = total + lov * self.loss_weights[self.loss_types.index("LovLoss")]
        return total
