"""`pcseg.loss.Losses` for the hot path (reference pcseg/loss/__init__.py:15-137).

TASeg's configs use CE (label smoothing, ignore 0) + Lovasz-softmax
(minkunet.py:344-348); those two are implemented.  The optional losses of the reference
(Dice, ELL, WCE, Focal, EQLv2, GroupSoftmax) are dense torch code outside the path and raise.
"""
import torch.nn as nn
from torch.nn import CrossEntropyLoss

from .lovasz import lovasz_softmax

__all__ = ["Losses", "lovasz_softmax"]

_SUPPORTED = ("CELoss", "LovLoss")
_OUT_OF_SCOPE = ("WCELoss", "ELLLoss", "DiceLossV0", "DiceLossV1", "FocalLoss", "EQLv2", "GroupSoftmax",
                 "GroupSoftmax_fgbg_2")


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
        self.lov_loss = lovasz_softmax

    def forward(self, input, target, xyz=None, offset=None):
        total = 0
        if "CELoss" in self.loss_types:
            total = total + self.ce_loss(input, target) * self.loss_weights[self.loss_types.index("CELoss")]
        if "LovLoss" in self.loss_types:
            lov = self.lov_loss(input.softmax(dim=1), target, ignore=self.ignore_index)
            total = total + lov * self.loss_weights[self.loss_types.index("LovLoss")]
        return total
