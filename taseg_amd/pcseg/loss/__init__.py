"""`pcseg.loss.Losses` for the hot path (reference pcseg/loss/__init__.py:15-137).

TASeg's configs use CE (label smoothing, ignore 0) + Lovasz-softmax
(minkunet.py:344-348); those two are implemented.  The optional losses of the reference
(Dice, ELL, WCE, Focal, EQLv2, GroupSoftmax) are dense torch code outside the path and raise.
"""
import os

import torch
import torch.nn as nn
from torch.nn import CrossEntropyLoss
from taseg_amd.options import options

from .lovasz import _NO_IGNORE, lovasz_softmax

_FUSED = options.fused_loss


class _CeLovasz(torch.autograd.Function):
    """w_ce * CrossEntropy(ignore_index, label_smoothing) + w_lov * lovasz_softmax(softmax(logits), ignore) on the HIP
    kernels of csrc/loss.hip: one pass over the logits (softmax, CE partial sums, Lovasz error matrix), torch's sort,
    ts_lovasz_grad, a one-workgroup finish; the backward pass is ONE kernel from the saved probabilities.  The tensor-op
    form (log_softmax, gathers, reductions, softmax and their autograd nodes) takes ~35 launches for the same numbers."""

    @staticmethod
    def forward(ctx, logits, target, ignore, smoothing, w_ce, w_lov):
        from ... import _lib as L
        lib = L.load()
        p, c = logits.shape
        x = logits.contiguous().float()
        lab = target.contiguous().long()
        dev = x.device
        ign = _NO_IGNORE if ignore is None else int(ignore)
        probas = torch.empty((p, c), dtype=torch.float32, device=dev)
        err = torch.empty((c, p), dtype=torch.float32, device=dev)
        partials = torch.empty(((p + 255) // 256, 3), dtype=torch.float64, device=dev)
        L.check(lib.ts_softmax_ce_forward(L.ptr(x), L.ptr(lab), ign, p, c, L.ptr(probas), L.ptr(err), L.ptr(partials),
                                          L.stream()), "ts_softmax_ce_forward")
        es, perm = torch.sort(err, dim=1, descending=True)
        lov = torch.empty(1, dtype=torch.float32, device=dev)
        dprob = torch.empty((c, p), dtype=torch.float32, device=dev)          # class-major: what the backward kernel reads
        ws = L.workspace(lib.ts_lovasz_workspace_bytes(p, c), dev)
        L.check(lib.ts_lovasz_grad(L.ptr(es), L.ptr(perm), L.ptr(lab), ign, p, c, L.ptr(lov), L.ptr(dprob), 1, L.ptr(ws),
                                   ws.numel(), L.stream()), "ts_lovasz_grad")
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        L.check(lib.ts_ce_lovasz_finish(L.ptr(partials), p, c, float(smoothing), float(w_ce), float(w_lov), L.ptr(lov),
                                        L.ptr(out4), L.stream()), "ts_ce_lovasz_finish")
        ctx.save_for_backward(probas, lab, dprob, out4)
        ctx.args = (ign, float(smoothing), float(w_ce), float(w_lov), logits.dtype)
        return out4[0]

    @staticmethod
    def backward(ctx, grad_out):
        from ... import _lib as L
        probas, lab, dprob, out4 = ctx.saved_tensors
        ign, smoothing, w_ce, w_lov, dtype = ctx.args
        p, c = probas.shape
        go = grad_out.contiguous().float().reshape(1)
        dlogits = torch.empty_like(probas)
        L.check(L.load().ts_ce_lovasz_backward(L.ptr(probas), L.ptr(lab), ign, L.ptr(dprob), L.ptr(out4), L.ptr(go), p, c,
                                               smoothing, w_ce, w_lov, L.ptr(dlogits), L.stream()), "ts_ce_lovasz_backward")
        return dlogits.to(dtype), None, None, None, None, None

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
        if (_FUSED and sorted(self.loss_types) == ["CELoss", "LovLoss"] and self.class_weight is None and input.dim() == 2
                and input.is_cuda and 0 < input.shape[1] <= 32 and input.shape[0] > 0
                and input.dtype in (torch.float32, torch.float16) and target.dim() == 1):
            return _CeLovasz.apply(input, target, self.ignore_index, self.label_smoothing,
                                   self.loss_weights[self.loss_types.index("CELoss")],
                                   self.loss_weights[self.loss_types.index("LovLoss")])
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
