"""Lovasz-softmax loss (Berman et al., CVPR 2018; the reference vendors the authors' MIT-licensed
PyTorch code at tools/utils/common/lovasz_losses.py:158-227).

Same value as the reference's per-class loop with classes='present', computed for all classes at
once: ONE batched descending sort of the [P, C] error matrix instead of C sorts, and no
`fg.sum() == 0` host synchronisation per class.
"""
import os

import torch
from taseg_amd.options import options

__all__ = ["lovasz_softmax", "lovasz_softmax_flat"]

_FUSED = options.fused_lovasz
_NO_IGNORE = -(1 << 62)          # a label value nothing takes


class _LovaszPresent(torch.autograd.Function):
    """classes = 'present' on the HIP kernels (csrc/loss.hip): the error matrix in one launch, torch's sort, then the
    prefix sums, the Lovasz gradient, the per-class dot products, the mean over the classes present AND the gradient
    w.r.t. the probabilities in four launches - the tensor-op form below takes ~40 launches forward and backward."""

    @staticmethod
    def forward(ctx, probas, labels, ignore):
        from ... import _lib as L
        lib = L.load()
        p, c = probas.shape
        prob = probas.contiguous().float()
        lab = labels.contiguous().long()
        ign = _NO_IGNORE if ignore is None else int(ignore)
        err = torch.empty((c, p), dtype=torch.float32, device=prob.device)
        L.check(lib.ts_lovasz_errors(L.ptr(prob), L.ptr(lab), ign, p, c, L.ptr(err), L.stream()), "ts_lovasz_errors")
        es, perm = torch.sort(err, dim=1, descending=True)
        loss = torch.empty(1, dtype=torch.float32, device=prob.device)
        dprob = torch.empty((p, c), dtype=torch.float32, device=prob.device)
        ws = L.workspace(lib.ts_lovasz_workspace_bytes(p, c), prob.device)
        L.check(lib.ts_lovasz_grad(L.ptr(es), L.ptr(perm), L.ptr(lab), ign, p, c, L.ptr(loss), L.ptr(dprob), 0, L.ptr(ws),
                                   ws.numel(), L.stream()), "ts_lovasz_grad")
        ctx.save_for_backward(dprob)
        ctx.in_dtype = probas.dtype
        return loss.view(())

    @staticmethod
    def backward(ctx, grad_out):
        (dprob,) = ctx.saved_tensors
        return (dprob * grad_out).to(ctx.in_dtype), None, None


def _cumsum_rows(x: torch.Tensor, block: int = 2048) -> torch.Tensor:
    """cumsum along dim 1 of a [C, P] tensor with few rows.  torch's innermost-dim scan walks each row with one
    workgroup (364 us for [20, 178k]); scanning [C * P/block, block] tiles plus a tiny scan of the tile totals
    keeps the whole device busy.  Same additions in the same order within a tile; exact for the 0/1 inputs here."""
    c, p = x.shape
    if p <= 4 * block:
        return x.cumsum(dim=1)
    pad = (-p) % block
    xp = torch.nn.functional.pad(x, (0, pad)) if pad else x
    tiles = xp.view(c, -1, block).cumsum(dim=2)
    totals = tiles[:, :, -1]
    base = totals.cumsum(dim=1) - totals                     # exclusive prefix of the tile totals
    return (tiles + base.unsqueeze(2)).view(c, -1)[:, :p]


def lovasz_softmax_flat(probas: torch.Tensor, labels: torch.Tensor, classes="present", valid=None) -> torch.Tensor:
    """probas [P, C] class probabilities, labels [P]; mean over classes of dot(sorted errors, Lovasz grad).

    `valid` (bool [P], optional) marks the rows that count.  Dropping the other rows (`probas[valid]`, what
    lovasz_losses.py:222-226 does) needs their number on the host - a device->host read that drains the whole
    forward pass from the launch queue right before backward.  Giving those rows zero foreground and zero error
    instead yields the same value: zero errors sort behind every positive error, where the Lovasz gradient is
    multiplied by 0, and the prefix sums in front of them (exact small integers in fp32) do not change."""
    if probas.numel() == 0:
        return probas * 0.0
    num_classes = probas.size(1)
    cls = torch.arange(num_classes, device=probas.device)
    # class-major [C, P] layout: sort / cumsum run along the contiguous last dimension (a cumsum over
    # dim 0 of a [P, C] tensor falls into a slow outer-dim scan kernel: 34 ms per call at P = 180k)
    hit = labels.view(1, -1) == cls.view(-1, 1)
    if valid is not None:
        hit = hit & valid.view(1, -1)
    fg = hit.to(probas.dtype)                                              # [C, P] one-hot foreground
    errors = (fg - probas.t()).abs()
    if valid is not None:
        errors = errors * valid.view(1, -1).to(probas.dtype)
    errors_sorted, perm = torch.sort(errors, dim=1, descending=True)
    fg_sorted = torch.gather(fg, 1, perm)
    # gradient of the Lovasz extension of the Jaccard loss w.r.t. sorted errors (Alg. 1)
    gts = fg_sorted.sum(dim=1, keepdim=True)
    # one scan instead of two: cumsum(1 - fg) = (position + 1) - cumsum(fg); all terms are integers < 2^24,
    # exact in fp32, so the values are those of the reference's two cumsums
    cum_fg = _cumsum_rows(fg_sorted)
    rank = torch.arange(1, fg_sorted.shape[1] + 1, device=fg_sorted.device, dtype=fg_sorted.dtype).view(1, -1)
    intersection = gts - cum_fg
    union = gts + (rank - cum_fg)
    jaccard = 1.0 - intersection / union
    grad = torch.cat([jaccard[:, :1], jaccard[:, 1:] - jaccard[:, :-1]], dim=1)
    per_class = (errors_sorted * grad).sum(dim=1)                          # [C]
    if classes == "all":
        return per_class.mean()
    if classes == "present":
        present = (gts.view(-1) > 0).to(per_class.dtype)
    else:
        present = torch.zeros_like(per_class)
        present[list(classes)] = 1.0
    # mean over the selected classes; 0 when none is selected (the reference's mean of an
    # empty list with its default `empty=0`)
    return (per_class * present).sum() / present.sum().clamp(min=1.0)


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None):
    """probas [P, C] (already flattened point-wise, as pcseg.loss passes them) and labels [P];
    rows whose label equals `ignore` are dropped first (lovasz_losses.py:207-227)."""
    if per_image:
        raise NotImplementedError("per_image Lovasz is an image-segmentation mode, unused by pcseg")
    if probas.dim() != 2:
        raise ValueError("lovasz_softmax here expects point-wise [P, C] probabilities")
    labels = labels.view(-1)
    if (_FUSED and classes == "present" and probas.is_cuda and probas.shape[0] > 0 and probas.shape[1] <= 64
            and probas.dtype in (torch.float32, torch.float16)):
        return _LovaszPresent.apply(probas, labels, ignore)
    valid = (labels != ignore) if ignore is not None else None
    return lovasz_softmax_flat(probas, labels, classes=classes, valid=valid)
