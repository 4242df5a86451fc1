"""Pre-split weight planes for the fp32 pair GEMMs (csrc/conv_pairs_s.hip, include/taseg_hip.h "Pre-split weight planes").

The fp32 convolution kernels run on the bf16 matrix pipe through the exact split x = h + m + l of both operands.  For a
weight that split is the same in every workgroup of every launch of a training step; `planes_for(weight)` keeps the six
bf16 planes of a convolution weight (h | m | l, each in the weight's own layout) next to it and re-splits them - all stale weights of
the model in one batch of launches - when the weight has changed: a different `_version` (torch optimizers,
`load_state_dict`, broadcasts), a different storage (`.to()`, FlatSGD's flat buckets) or an `invalidate()` from code
that writes parameters behind torch's back (FlatSGD's ts_sgd_apply).  The planes are handed to the backend as a
one-shot hint per call (ts_conv_planes_hint); results are bit-identical with and without them.

TASEG_PRESPLIT=0 switches the mechanism off.
"""
import ctypes
import os
import weakref

import torch

from . import _lib as L
from .options import options

__all__ = ["planes_for", "half_for", "invalidate", "eligible", "eligible_half", "hint", "stats"]

_ENABLED = options.presplit
_entries = {}          # id(weight) -> _Entry (bf16 planes of the fp32 kernels)
_half_entries = {}     # id(weight) -> _Entry (IEEE-half copy for the half-storage kernels)
_epoch = 0
stats = {"refreshes": 0, "launch_batches": 0}


class TsPlaneJob(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("planes", ctypes.c_void_p), ("K", ctypes.c_int32), ("c_in", ctypes.c_int32),
                ("c_out", ctypes.c_int32)]


class _Entry:
    __slots__ = ("ref", "planes", "ptr", "version", "epoch", "stream")

    def __init__(self, weight, half=False):
        self.ref = weakref.ref(weight)
        self.planes = (torch.empty(weight.shape, dtype=torch.float16, device=weight.device) if half else
                       torch.empty(3 * weight.numel(), dtype=torch.int16, device=weight.device))
        self.ptr = self.version = self.epoch = self.stream = None

    def fresh(self, weight, stream):
        return self.ptr == weight.data_ptr() and self.version == weight._version and self.epoch == _epoch \
            and self.stream == stream


def invalidate():
    """Every weight may have changed without a version bump (a kernel wrote the parameters through raw pointers)."""
    global _epoch
    _epoch += 1


def _wide(c):
    return c % 128 == 0 and c % 96 != 0      # the widths the direct-rows kernel takes (128-column tiles)


def eligible(weight) -> bool:
    """fp32 [K, C_in, C_out] weight on a ROCm device whose forward product or input gradient runs on 128-column tiles."""
    return (_ENABLED and weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 3 and weight.is_contiguous()
            and weight.shape[0] <= 63 and weight.shape[1] % 32 == 0 and weight.shape[2] % 32 == 0
            and (_wide(weight.shape[1]) or _wide(weight.shape[2])))


def _refresh(stream, device, entries=None, fn="ts_conv_split_planes_batch"):
    """Re-split every registered weight ON `device` that is stale (one launch per 16 weights) on `stream` - a stream of
    that device: the caller's current one.  Weights of a second model on another GPU of the same process stay stale until
    a convolution on their own device asks for them."""
    entries = _entries if entries is None else entries
    jobs, done = [], []
    for key, e in list(entries.items()):
        w = e.ref()
        if w is None:
            del entries[key]
            continue
        if w.device != device:
            continue
        if not e.fresh(w, stream):
            k, c_in, c_out = w.shape if w.dim() == 3 else (1, w.shape[0], w.shape[1])      # (1x1x1 weights are [C_in, C_out])
            jobs.append((w.data_ptr(), e.planes.data_ptr(), k, c_in, c_out))
            done.append((e, w))
    if not jobs:
        return
    arr = (TsPlaneJob * len(jobs))(*[TsPlaneJob(*j) for j in jobs])
    L.check(getattr(L.load(), fn)(arr, len(jobs), stream), fn)
    for e, w in done:
        e.ptr, e.version, e.epoch, e.stream = w.data_ptr(), w._version, _epoch, stream
    stats["refreshes"] += len(jobs)
    stats["launch_batches"] += 1


def planes_for(weight):
    """The planes tensor of `weight` (int16 storage of 3 * numel bf16), in step with the weight on the current stream,
    or None when the mechanism does not apply."""
    e = _entries.get(id(weight))
    if e is not None and e.ref() is weight and e.fresh(weight, L.stream()) and e.planes.device == weight.device:
        return e.planes              # (the common case first: same storage, version, epoch and stream as when the planes were made)
    if not eligible(weight):
        return None
    if weight.device.index != torch.cuda.current_device():
        # the split runs on the caller's current stream, which belongs to the current device: a weight that lives on
        # another GPU goes without planes (the kernels then split in the workgroups, same bits)
        return None
    stream = L.stream()
    e = _entries.get(id(weight))
    if e is None or e.ref() is not weight or e.planes.device != weight.device:
        e = _entries[id(weight)] = _Entry(weight)
    if not e.fresh(weight, stream):
        _refresh(stream, weight.device)
    return e.planes


def eligible_half(weight) -> bool:
    """fp32 [K, C_in, C_out] (1x1x1: [C_in, C_out]) parameter on a ROCm device that the half-storage block call can take a kept
    copy of."""
    return (_ENABLED and weight.is_cuda and weight.dtype == torch.float32 and weight.dim() in (2, 3) and weight.is_contiguous()
            and (weight.shape[-2] * weight.shape[-1]) % 8 == 0 and weight.data_ptr() % 16 == 0)


def half_for(weight):
    """The IEEE-half copy [K, C_in, C_out] of `weight` that the half-storage convolutions read (autocast's cast of the
    weight, conv.py:19, done once per optimizer step for ALL weights of the model - 16 per launch - instead of once per
    convolution call), in step with the weight on the current stream; None when the mechanism does not apply.  Hand it to
    the block call as its w16 buffer and name it in `hint(weight, w16)`: the call then launches no cast."""
    e = _half_entries.get(id(weight))
    if e is not None and e.ref() is weight and e.fresh(weight, L.stream()) and e.planes.device == weight.device \
            and e.planes.shape == weight.shape:
        return e.planes              # (the common case first)
    if not eligible_half(weight) or weight.device.index != torch.cuda.current_device():
        return None
    stream = L.stream()
    e = _half_entries.get(id(weight))
    if e is None or e.ref() is not weight or e.planes.device != weight.device or e.planes.shape != weight.shape:
        e = _half_entries[id(weight)] = _Entry(weight, half=True)
    if not e.fresh(weight, stream):
        _refresh(stream, weight.device, _half_entries, "ts_cast_weights_f16_batch")
    return e.planes


def hint(weight32, planes):
    """One-shot: the next ts_conv_pair_gemm / ts_conv_block_* call of this thread on `weight32` may read `planes`."""
    if planes is not None:
        k, c_in, c_out = weight32.shape
        L.load().ts_conv_planes_hint(weight32.data_ptr(), planes.data_ptr(), k, c_in, c_out)
