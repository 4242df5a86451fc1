"""Flat-bucket SGD for the training step (reference R/train.py:399-417, R/pcseg/optim/__init__.py:13-21).

`FlatSGD` keeps parameters, gradients and momentum in a few flat fp32 buckets and performs
`GradScaler.unscale_ -> clip_grad_norm_ -> SGD.step -> GradScaler.update` with three kinds of HIP launches per
step (ts_sgd_grad_stats per bucket, ts_sgd_decide once, ts_sgd_apply per bucket) and NO device->host read: the skip
decision on non-finite gradients, the clip coefficient and the loss-scale schedule live on the device.  torch's
equivalent is ~10 multi-tensor launches over 380 tensors plus `found_inf.item()`, a queue-draining read per step.

Gradients arrive through `parallel.GradBucketReducer` (one multi-tensor copy per bucket during backward, then the
all-reduce over ranks when there are several), whose buckets this optimizer shares.

    opt = FlatSGD(model, lr=..., momentum=0.9, weight_decay=1e-4, max_norm=10.0, amp=True)
    with torch.autocast("cuda", dtype=torch.float16): loss = ...
    (loss * opt.loss_scale()).backward()        # device scalar, no sync
    opt.step()                                  # reducer.finish() + the three launches
"""
import torch

from . import _lib as L
from . import planes as _planes
from .parallel import GradBucketReducer

__all__ = ["FlatSGD"]


class FlatSGD:
    def __init__(self, model: torch.nn.Module, lr: float, momentum: float = 0.9, weight_decay: float = 0.0,
                 max_norm: float = 0.0, amp: bool = False, init_scale: float = 65536.0, growth_factor: float = 2.0,
                 backoff_factor: float = 0.5, growth_interval: int = 2000, process_group=None, bucket_mb: float = 32.0):
        self.lr, self.momentum, self.weight_decay, self.max_norm = lr, momentum, weight_decay, max_norm
        self.amp, self.growth, self.backoff, self.interval = amp, growth_factor, backoff_factor, growth_interval
        self.reducer = GradBucketReducer(model, process_group=process_group, bucket_mb=bucket_mb)
        dev = self.reducer.buckets[0]["flat"].device
        L.require_device(self.reducer.buckets[0]["flat"])
        # parameters move into flat buckets too (p.data becomes a view): the update is one launch per bucket
        for b in self.reducer.buckets:
            flat = torch.zeros_like(b["flat"])
            for p, off in zip(b["params"], b["offsets"]):      # same (64-byte aligned) layout as the gradient bucket
                if p.dtype != torch.float32:
                    raise TypeError("FlatSGD keeps fp32 master parameters")
                view = flat[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
            b["pflat"], b["mflat"] = flat, torch.zeros_like(flat)
        self.state = torch.zeros(8, dtype=torch.float32, device=dev)
        self.state[0] = init_scale if amp else 1.0
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self._bad = torch.zeros(1, dtype=torch.int32, device=dev)
        self._first = True

    def loss_scale(self) -> torch.Tensor:
        """Device scalar to multiply the loss with before backward (1 without AMP)."""
        return self.state[0]

    def zero_grad(self, set_to_none: bool = True):
        from .parallel import bump_grad_epoch
        bump_grad_epoch()
        for b in self.reducer.buckets:
            for p in b["params"]:
                p.grad = None
                p._taseg_dest_claimed = False

    def step(self):
        self.reducer.finish()                         # gradients sit in the flat buckets (reduced over ranks)
        lib, st = L.load(), L.stream()
        for b in self.reducer.buckets:
            L.check(lib.ts_sgd_grad_stats(L.ptr(b["flat"]), b["flat"].numel(), L.ptr(self._sumsq), L.ptr(self._bad), st),
                    "ts_sgd_grad_stats")
        L.check(lib.ts_sgd_decide(L.ptr(self._sumsq), L.ptr(self._bad), L.ptr(self.state), float(self.max_norm),
                                  float(self.growth), float(self.backoff), int(self.interval), 1 if self.amp else 0, st),
                "ts_sgd_decide")
        for b in self.reducer.buckets:
            # parameters that received no gradient on ANY rank this step (an unused head, a frozen branch): torch.optim.SGD
            # skips them entirely - no weight decay, no momentum update (R/pcseg/optim/__init__.py:15-21 builds that
            # optimizer).  The flat update touches the whole bucket, so their slices are put back afterwards.  b["unused"]
            # is this rank's list; with several ranks the reducer's all-reduced flag of the parameter says whether some
            # OTHER rank had a gradient for it - then the averaged gradient is applied here too (DDP's rule) and the
            # replicas stay identical.  The flag is read on the device (torch.where), not on the host.
            keep = []
            for i in b.get("unused", ()):
                off, n = b["offsets"][i], b["params"][i].numel()
                used_elsewhere = (b["flags"][i] > 0) if self.reducer.world > 1 else None
                keep.append((off, n, b["pflat"][off:off + n].clone(), b["mflat"][off:off + n].clone(), used_elsewhere))
            L.check(lib.ts_sgd_apply(L.ptr(b["pflat"]), L.ptr(b["flat"]), L.ptr(b["mflat"]), b["flat"].numel(),
                                     L.ptr(self.state), float(self.lr), float(self.momentum), float(self.weight_decay),
                                     1 if self._first else 0, st), "ts_sgd_apply")
            for off, n, pv, mv, used_elsewhere in keep:
                if used_elsewhere is None:
                    b["pflat"][off:off + n].copy_(pv)
                    b["mflat"][off:off + n].copy_(mv)
                else:
                    b["pflat"][off:off + n].copy_(torch.where(used_elsewhere, b["pflat"][off:off + n], pv))
                    b["mflat"][off:off + n].copy_(torch.where(used_elsewhere, b["mflat"][off:off + n], mv))
        # (a skipped very first step leaves the momentum buffers zero, which `first` = 0 then treats correctly:
        #  momentum * 0 + d = d)
        self._first = False
        _planes.invalidate()       # ts_sgd_apply wrote the parameters through raw pointers: pre-split weights are stale
