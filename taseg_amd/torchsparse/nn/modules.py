"""`torchsparse.nn` modules (TS/torchsparse/nn/modules/{conv,norm,activation}.py)."""
import math
import os

import torch
from torch import nn

from ..tensor import SparseTensor
from ..utils.misc import make_ntuple
from . import functional as F
from ... import backend as _B
from ... import _fast
from ... import planes as _planes
from ...options import options
from .utils import fapply

__all__ = ["Conv3d", "BatchNorm", "SyncBatchNorm", "ReLU", "LeakyReLU", "bn_act", "conv_bn_act"]


_FUSED_BLOCK = options.fused_block
# the 1x1x1 shortcut + its BatchNorm as one block call on the identity rulebook (_pointwise_bn_act); 0: GEMM node + BatchNorm node
_POINTWISE_BLOCK = options.pointwise_block
_DIL1 = (1, 1, 1)


class Conv3d(nn.Module):
    """Sparse 3-D convolution.  Parameter ``kernel`` is [K, C_in, C_out] (K = kernel volume,
    indexed by `get_kernel_offsets` order) or [C_in, C_out] for a 1x1x1 kernel - the same
    state-dict layout as the reference (modules/conv.py:33-38), init U(+-1/sqrt(fan * K))."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size=3, stride=1, dilation: int = 1,
                 bias: bool = False, transposed: bool = False) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = make_ntuple(kernel_size, ndim=3)
        self.stride = make_ntuple(stride, ndim=3)
        self.dilation = dilation
        self.transposed = transposed
        self.kernel_volume = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        shape = (self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.zeros(*shape))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        fan = self.out_channels if self.transposed else self.in_channels
        bound = 1.0 / math.sqrt(fan * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def extra_repr(self) -> str:
        parts = [f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}"]
        if self.stride != (1, 1, 1):
            parts.append(f"stride={self.stride}")
        if self.dilation != 1:
            parts.append(f"dilation={self.dilation}")
        if self.bias is None:
            parts.append("bias=False")
        if self.transposed:
            parts.append("transposed=True")
        return ", ".join(parts)

    def forward(self, input: SparseTensor) -> SparseTensor:
        return F.conv3d(input, self.kernel, kernel_size=self.kernel_size, bias=self.bias, stride=self.stride,
                        dilation=self.dilation, transposed=self.transposed,
                        planes=None if F._amp_half(input.feats) else _planes.planes_for(self.kernel))


def _bn_forward(mod, feats, torch_forward, group=None):
    """Training mode with batch statistics -> our reduction kernels; everything else (eval mode,
    momentum=None cumulative averaging, odd channel counts) -> the stock torch module."""
    from .batchnorm import batch_norm_train, fast_path_ok
    _require_rows(feats, group)
    if mod.training and mod.momentum is not None and mod.affine and fast_path_ok(feats):
        rm = mod.running_mean if mod.track_running_stats else None
        rv = mod.running_var if mod.track_running_stats else None
        nbt = mod.num_batches_tracked if mod.track_running_stats else None
        return batch_norm_train(feats, mod.weight, mod.bias, rm, rv, mod.momentum, mod.eps, group,
                                num_batches_tracked=nbt)
    return torch_forward(feats)


def _sync_group(mod):
    """process group whose ranks share batch statistics, or None for local statistics.  A one-rank group behaves
    like None; TASEG_SYNCBN_SINGLE_RANK=1 keeps the collective path on anyway (lets one GPU exercise it)."""
    if not isinstance(mod, nn.SyncBatchNorm):           # (first: 63 calls per pass, and a plain BatchNorm needs none of the rest)
        return None
    import os
    import torch.distributed as dist
    if mod.training and dist.is_available() and dist.is_initialized() \
            and (dist.get_world_size() > 1 or options.syncbn_single_rank):
        return mod.process_group if mod.process_group is not None else dist.group.WORLD
    return None


def _require_rows(feats, group):
    """Which collectives a SyncBatchNorm issues must not depend on rank-local data (a rank taking another transport
    than its peers hangs the job).  Every shape rule below is a function of channel count and dtype only - except an
    EMPTY shard, which no kernel path serves: the path shards whole scans, so a rank without voxels is a data-loading
    error and is reported as one instead of a hang."""
    if group is not None and feats.shape[0] == 0:
        raise RuntimeError("SyncBatchNorm: this rank holds no voxels (every rank must receive at least one scan)")


def bn_act(mod, input: SparseTensor, relu: bool = True, residual: SparseTensor = None) -> SparseTensor:
    """relu(BN(input) [+ residual]) for a BatchNorm / SyncBatchNorm module `mod` - the tail of every conv block of
    the MinkUNet family (minkunet.py:42-51, 117-129).  Training mode runs as two fused passes on the HIP
    kernels; otherwise the stock modules are chained exactly like the reference does."""
    from .batchnorm import batch_norm_act_train, fast_path_ok
    feats = input.feats
    res = None if residual is None else residual.feats
    _require_rows(feats, _sync_group(mod))
    if mod.training and mod.momentum is not None and mod.affine and fast_path_ok(feats) \
            and (res is None or (res.shape == feats.shape and res.dtype == feats.dtype)):
        rm = mod.running_mean if mod.track_running_stats else None
        rv = mod.running_var if mod.track_running_stats else None
        nbt = mod.num_batches_tracked if mod.track_running_stats else None
        out = batch_norm_act_train(feats, mod.weight, mod.bias, rm, rv, mod.momentum, mod.eps, relu=relu,
                                   residual=res, group=_sync_group(mod), num_batches_tracked=nbt)
        return input._like(out)
    out = mod(input).feats
    if res is not None:
        out = out + res
    if relu:
        out = torch.relu(out)
    return input._like(out)


_group_ids = {}


def _group_id(fast, group):
    """index of a torch.distributed process group in the native node's registry (-1 = no c10d all-reduce in the block)"""
    if group is None:
        return -1
    key = id(group)
    if key not in _group_ids:
        _group_ids[key] = fast.register_group(group)
    return _group_ids[key]


def _claim_grad_dest(kernel):
    """The gradient-bucket slot of `kernel` (parallel.GradBucketReducer) as the place the block's backward may write the
    weight gradient STRAIGHT into - or None.  The slot is a full-overwrite target that autograd then adopts as p.grad, so it
    is handed out only while nothing can have been accumulated for this step: p.grad is None (zero_grad(set_to_none=True)
    semantics; a kept .grad aliases the slot itself and an overwrite followed by `grad += alias` would double the new
    value) and no other use of the same weight since the last reducer.finish() / zero_grad() has claimed it (a weight
    used twice in one graph: the second node gets a fresh tensor and autograd adds the two).  The claim is dropped by
    GradBucketReducer.finish() and FlatSGD.zero_grad()."""
    dest = getattr(kernel, "_taseg_grad_dest", None)
    if dest is None or kernel.grad is not None or getattr(kernel, "_taseg_dest_claimed", False):
        return None
    kernel._taseg_dest_claimed = True
    return dest


_NO_PLAN = ([], [])


def _plan_args(plan):
    """(tensors, meta) of a class plan for the C++ node: (src, tile_info, n_tiles, pos | rows) and (n, m_pad, z_rows, K, groups,
    mirror, direct) - or two empty lists"""
    if plan is None:
        return _NO_PLAN
    hit = plan.get("_args")
    if hit is None:
        direct = plan["rows"] is not None
        hit = ([plan["src"], plan["tile_info"], plan["n_tiles"], plan["rows"] if direct else plan["pos"]],
               [plan["n"], plan["m_pad"], int(plan.get("z_rows") or 0), plan["K"], plan["groups"], plan["mirror"], 1 if direct else 0])
        plan["_args"] = hit
    return hit


def _eval_invstd(mod):
    """1 / sqrt(running_var + eps) of a BatchNorm module, kept until the buffer changes (63 modules x 10 TTA votes per scan)"""
    rv = mod._buffers["running_var"]
    hit = mod.__dict__.get("_taseg_eval_invstd")
    if hit is None or hit[0] != rv._version or hit[1] is not rv or hit[2].device != rv.device:
        with torch.no_grad():
            hit = (rv._version, rv, torch.rsqrt(rv.float() + mod.eps))
        mod._taseg_eval_invstd = hit
    return hit[2]


def _pointwise_bn_act(conv, mod, input: SparseTensor, relu, residual):
    """relu(BN(conv(input)) [+ residual]) for a 1x1x1 convolution (conv.py:135-140: a dense GEMM over the rows) as one native block
    call per direction on the identity rulebook - the shortcut of a residual block (minkunet.py:105-111); None when the block
    call does not apply (the caller then chains the modules)."""
    ones = (1, 1, 1)
    fast = _fast.module()
    if conv.bias is not None or make_ntuple(conv.stride, ndim=3) != ones \
            or make_ntuple(conv.dilation, ndim=3) != ones or conv.transposed or not mod.affine \
            or conv._forward_hooks or conv._forward_pre_hooks:
        return None
    feats = input.feats
    res = None if residual is None else residual.feats
    if not F.pointwise_block_ok(feats, conv.kernel, res) or mod.weight.dtype != torch.float32:
        return None
    half = F._amp_half(feats)
    n = feats.shape[0]
    imap = F.identity_map(n, feats.device)
    w16 = _planes.half_for(conv.kernel) if half else None
    if not mod.training and not torch.is_grad_enabled() and mod.track_running_stats and mod.running_var is not None \
            and mod.running_mean.dtype == torch.float32 and not (mod._forward_hooks or mod._forward_pre_hooks):
        if fast is not None:
            out = fast.conv_block_eval(feats, conv.kernel, res, mod.weight, mod.bias, mod.running_mean, _eval_invstd(mod),
                                       imap.nbmaps_buf, imap.nboffs, n, imap.pos_out, imap.pos_in, n, n, False, relu, half,
                                       _B.L.stream(), w16, [], [], True)
        else:
            out = F.conv_block_eval(feats, conv.kernel, res, mod.weight, mod.bias, mod.running_mean, _eval_invstd(mod), imap,
                                    False, relu, half, w16)
        return input._like(out)
    if not (mod.training and torch.is_grad_enabled() and mod.momentum is not None):
        return None
    from ...rccl import direct_comm
    group = _sync_group(mod)
    _require_rows(feats, group)
    comm = None if group is None else direct_comm(group)
    track = mod.track_running_stats
    if fast is None:
        state = (mod.running_mean if track else None, mod.running_var if track else None,
                 mod.num_batches_tracked if track else None, mod.momentum, mod.eps)
        out = F._ConvBlock.apply(feats, conv.kernel, res, mod.weight, mod.bias, imap, False, state, relu, comm, half, w16, False,
                                 _claim_grad_dest(conv.kernel), group if (group is not None and comm is None) else None)
        return input._like(out)
    out = fast.conv_block(feats, conv.kernel, res, mod.weight, mod.bias, imap.nbmaps_buf, imap.nboffs, n, imap.pos_out,
                          imap.pos_in, n, n, False, mod.running_mean if track else None, mod.running_var if track else None,
                          mod.num_batches_tracked if track else None, float(mod.momentum), float(mod.eps), relu,
                          (comm.value or 0) if comm is not None else 0, half, _B.L.stream(), w16, False,
                          _claim_grad_dest(conv.kernel), _group_id(fast, group if (group is not None and comm is None) else None),
                          [], [], [], [], conv.kernel.grad is None, True)
    return input._like(out[0])


def conv_bn_act(conv: "Conv3d", mod, input: SparseTensor, relu: bool = True, residual: SparseTensor = None,
                passthrough: bool = False):
    """relu(BN(conv(input)) [+ residual]) for a Conv3d and its BatchNorm / SyncBatchNorm module: one autograd node and
    one backend call per direction when the block trains on the HIP path (functional._ConvBlock); otherwise exactly
    `bn_act(mod, conv(input), relu, residual)`.  TASEG_FUSED_BLOCK=0 always takes the second form.

    passthrough=True returns (output, input'): input' carries the input's features through the node, so that a second
    consumer of the input (the shortcut of a residual block) sends its gradient back INTO the node, where it is added
    in the store of the convolution's input gradient instead of by a separate add launch.  Use input' in place of
    `input` downstream; on the unfused paths input' is `input` itself."""
    ones = (1, 1, 1)
    ks, stride = conv.kernel_size, conv.stride
    dil = conv.dilation
    if type(dil) is not tuple or len(dil) != 3:
        dil = _DIL1 if dil == 1 else make_ntuple(dil, ndim=3)
    if ks == ones:
        out = _pointwise_bn_act(conv, mod, input, relu, residual) if (_FUSED_BLOCK and _POINTWISE_BLOCK and not passthrough) else None
        if out is None:
            out = bn_act(mod, conv(input), relu=relu, residual=residual)
        return (out, input) if passthrough else out
    # (parameters and buffers are read from the modules' own dictionaries: `conv.kernel` / `mod.weight` go through
    # nn.Module.__getattr__ after a failed instance lookup, ~0.5 us each and ~15 of them per block call)
    cpar, mpar, mbuf = conv._parameters, mod._parameters, mod._buffers
    if _FUSED_BLOCK and cpar["bias"] is None and not mod.training and not torch.is_grad_enabled() and mod.affine \
            and mod.track_running_stats and mbuf.get("running_var") is not None \
            and not (conv._forward_hooks or conv._forward_pre_hooks or mod._forward_hooks or mod._forward_pre_hooks):
        # evaluation (eval-mode BatchNorm, no graph): one backend call per block on the running statistics
        kernel, transposed = cpar["kernel"], conv.transposed
        kmap, out_coords, out_stride = F.conv_geometry(input, ks, stride, dil, transposed)
        n_in, n_out = kmap.sizes
        rows = n_in if transposed else n_out
        res = None if residual is None else residual.feats
        feats = input.feats
        if feats.shape[0] == (n_out if transposed else n_in) and F.conv_block_ok(feats, kernel, kmap, res, rows):
            half = F._amp_half(feats)
            planes = _planes.half_for(kernel) if half else _planes.planes_for(kernel)
            fast = _fast.module()
            bn_w, bn_b, rmean = mpar["weight"], mpar["bias"], mbuf["running_mean"]
            if fast is not None and rmean.dtype == torch.float32 and bn_w.dtype == torch.float32:
                plan_f, _ = kmap.plans_for(transposed, kernel.shape[1], kernel.shape[2], half)
                out = fast.conv_block_eval(feats, kernel, res, bn_w, bn_b, rmean, _eval_invstd(mod),
                                           kmap.nbmaps_buf, kmap.nboffs, kmap.total, kmap.pos_out, kmap.pos_in, n_in, n_out,
                                           transposed, relu, half, _B.L.stream(), planes, *_plan_args(plan_f), False)
            else:
                out = F.conv_block_eval(feats, kernel, res, bn_w, bn_b, rmean, _eval_invstd(mod), kmap,
                                        transposed, relu, half, planes)
            result = F._conv_output(input, out, out_coords, out_stride)
            return (result, input) if passthrough else result
    if _FUSED_BLOCK and cpar["bias"] is None and mod.training and torch.is_grad_enabled() and mod.momentum is not None \
            and mod.affine and not (conv._forward_hooks or conv._forward_pre_hooks):      # hooks on the conv module must still fire
        group = _sync_group(mod)
        comm = None
        if group is not None:
            from ...rccl import direct_comm
            _require_rows(input.feats, group)
            comm = direct_comm(group)
        # SyncBatchNorm without the library-owned communicator: the Python node splits the block call around c10d's all-reduce
        # (module attributes are read once: every `conv.kernel` / `mod.weight` is a trip through nn.Module.__getattr__, ~1000 of
        # them per pass made 0.2 ms of a host-bound step)
        kernel, transposed = cpar["kernel"], conv.transposed
        kmap, out_coords, out_stride = F.conv_geometry(input, ks, stride, dil, transposed)
        n_in, n_out = kmap.sizes
        rows = n_in if transposed else n_out
        res = None if residual is None else residual.feats
        feats = input.feats
        if feats.shape[0] == (n_out if transposed else n_in) and F.conv_block_ok(feats, kernel, kmap, res, rows):
            track = mod.track_running_stats
            momentum, eps = mod.momentum, mod.eps
            state = (mbuf["running_mean"] if track else None, mbuf["running_var"] if track else None,
                     mbuf["num_batches_tracked"] if track else None, momentum, eps)
            bn_w, bn_b = mpar["weight"], mpar["bias"]
            fast = _fast.module()
            half = F._amp_half(feats)
            # fp32: pre-split bf16 planes of the weight; half storage: its kept half copy (taseg_amd/planes.py)
            planes = _planes.half_for(kernel) if half else _planes.planes_for(kernel)
            dest = _claim_grad_dest(kernel)                            # bucket slot of the weight gradient (parallel.py)
            c10d_group = group if (group is not None and comm is None) else None
            # class plans of this kernel map for the forward product / the input gradient (functional.KernelMap.plans_for)
            plan_f, plan_d = kmap.plans_for(transposed, kernel.shape[1], kernel.shape[2], half)
            if fast is not None:                          # C++ autograd node, same backend calls (csrc/fastpath)
                out = fast.conv_block(feats, kernel, res, bn_w, bn_b, kmap.nbmaps_buf, kmap.nboffs,
                                      kmap.total, kmap.pos_out, kmap.pos_in, n_in, n_out, transposed, state[0],
                                      state[1], state[2], float(momentum), float(eps), relu,
                                      (comm.value or 0) if comm is not None else 0, half, _B.L.stream(), planes,
                                      bool(passthrough), dest, _group_id(fast, c10d_group), *_plan_args(plan_f), *_plan_args(plan_d),
                                      kernel.grad is None, False)
                out, passed = (out[0], out[1]) if passthrough else (out[0], None)
            else:
                out = F._ConvBlock.apply(feats, kernel, res, bn_w, bn_b, kmap, transposed, state,
                                         relu, comm, half, planes, bool(passthrough), dest, c10d_group)
                out, passed = out if passthrough else (out, None)
            result = F._conv_output(input, out, out_coords, out_stride)
            return (result, input._like(passed)) if passthrough else result
    result = bn_act(mod, conv(input), relu=relu, residual=residual)
    return (result, input) if passthrough else result


class BatchNorm(nn.BatchNorm1d):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, lambda f: _bn_forward(self, f, super(BatchNorm, self).forward))


class SyncBatchNorm(nn.SyncBatchNorm):
    """nn.SyncBatchNorm over sparse features: statistics over all ranks (one all-reduce of [2C+1] doubles)."""

    def forward(self, input: SparseTensor) -> SparseTensor:
        group = _sync_group(self)
        return fapply(input, lambda f: _bn_forward(self, f, super(SyncBatchNorm, self).forward, group))


class ReLU(nn.ReLU):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class PointLinear(nn.Linear):
    """nn.Linear over [N, C] point features (same parameters / state_dict); training on a ROCm device routes the
    weight gradient through the HIP reduction kernel (functional.point_linear)."""

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return F.point_linear(input, self.weight, self.bias)
