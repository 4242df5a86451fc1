"""Training-mode BatchNorm (+ residual) (+ ReLU) over sparse-tensor features on the HIP kernels of csrc/bn.hip.
Semantics = nn.BatchNorm1d / nn.SyncBatchNorm in training mode followed by the residual add and ReLU of the MinkUNet
blocks (reference: minkunet.py:23-29 wraps the torch modules with fapply, :42-51 / :110-129 chain the passes).
Single process: one backend call per direction (`ts_bn_act_train_*`, fp32 or half storage).  SyncBatchNorm: local
sliced sums -> ONE all-reduce of [2C+1] (forward) / [2C] (backward) doubles -> elementwise kernels; torch's own
SyncBatchNorm (torch/nn/modules/_functions.py) all-gathers per-rank mean / invstd / count instead."""
import torch
from torch.autograd import Function

from ... import _lib as L
from ...rccl import c10d_sum, direct_comm

__all__ = ["batch_norm_train", "fast_path_ok"]


def fast_path_ok(x: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and x.shape[1] <= 1024):
        return False
    return (x.dtype == torch.float32 and x.shape[1] % 4 == 0) or (x.dtype == torch.float16 and x.shape[1] % 8 == 0)


def _bump(*buffers):
    """the kernels wrote these buffers through raw pointers: move their version counters, as an in-place torch op would have -
    whoever caches something derived from them (the evaluation blocks' 1 / sqrt(running_var + eps)) keys on the version"""
    for b in buffers:
        if b is not None:
            torch.autograd.graph.increment_version(b)


class _BatchNormActTrain(Function):
    """act(BN(x) [+ residual]) in training mode, forward and backward each as (one reduction + one elementwise
    pass) over [N, C]; the ReLU mask is one bit per element written by the forward."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, nbt, momentum, eps, relu, group):
        half = x.dtype == torch.float16
        x = x.contiguous() if (half or x.dtype == torch.float32) else x.contiguous().float()
        n, c = x.shape
        lib = L.load()
        sfx = "_f16" if half else ""
        if residual is not None:
            residual = residual.contiguous().to(x.dtype)
        stats = torch.empty((2, c), dtype=torch.float32, device=x.device)        # mean, invstd
        mean, invstd = stats[0], stats[1]
        out = torch.empty_like(x)
        per = 8 if half else 4                         # elements per mask byte
        mask = torch.empty(n * (c // per), dtype=torch.uint8, device=x.device) if relu else None
        total_dev = None
        ws = L.workspace(lib.ts_bn_train_workspace_bytes(c), x.device)
        comm = None if group is None else direct_comm(group)
        if group is None:
            # single process: partial reductions, statistics and the elementwise pass in one backend call
            # (half storage: activations half in / out, statistics and arithmetic fp32 - what autocast does to batch_norm)
            fn = getattr(lib, "ts_bn_act_train_forward" + sfx)
            L.check(fn(L.ptr(x), L.ptr(residual), L.ptr(weight), L.ptr(bias), L.ptr(running_mean), L.ptr(running_var),
                       L.ptr(nbt), n, c, float(eps), float(momentum), 1 if relu else 0, L.ptr(mean), L.ptr(invstd),
                       L.ptr(out), L.ptr(mask), L.ptr(ws), ws.numel(), L.stream()), "ts_bn_act_train_forward" + sfx)
        elif comm is not None:
            # SyncBatchNorm, library-owned communicator: local sums -> ncclAllReduce([2C + 1] doubles) -> statistics ->
            # elementwise pass, ONE backend call, everything in order on this stream (csrc/rccl.hip)
            pack = torch.empty(2 * c + 1, dtype=torch.float64, device=x.device)
            L.check(lib.ts_bn_sync_forward(comm, L.ptr(x), L.ptr(residual), L.ptr(weight), L.ptr(bias),
                                           L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt), n, c, float(eps),
                                           float(momentum), 1 if relu else 0, 1 if half else 0, L.ptr(pack), L.ptr(mean),
                                           L.ptr(invstd), L.ptr(out), L.ptr(mask), L.ptr(ws), ws.numel(), L.stream()),
                    "ts_bn_sync_forward")
            total_dev = pack[2 * c:]
        else:
            # SyncBatchNorm through the process group: local sums -> ONE all-reduce of [2C + 1] doubles -> statistics +
            # elementwise pass
            if nbt is not None:
                nbt.add_(1)
            pack = torch.empty(2 * c + 1, dtype=torch.float64, device=x.device)
            L.check(getattr(lib, "ts_bn_sync_stats" + sfx)(L.ptr(x), n, c, L.ptr(pack), L.ptr(ws), ws.numel(),
                                                           L.stream()), "ts_bn_sync_stats" + sfx)
            c10d_sum(pack, group)
            total_dev = pack[2 * c:]
            L.check(lib.ts_bn_finalize(L.ptr(pack), L.ptr(total_dev), float(n), c, float(eps), float(momentum),
                                       L.ptr(running_mean), L.ptr(running_var), L.ptr(mean), L.ptr(invstd),
                                       L.stream()), "ts_bn_finalize")
            L.check(getattr(lib, "ts_bn_act_forward" + sfx)(L.ptr(x), L.ptr(residual), L.ptr(mean), L.ptr(invstd),
                                                            L.ptr(weight), L.ptr(bias), n, c, 1 if relu else 0,
                                                            L.ptr(out), L.ptr(mask), L.stream()),
                    "ts_bn_act_forward" + sfx)
        ctx.save_for_backward(x, weight, mean, invstd, mask)
        ctx.group, ctx.total_dev, ctx.has_res, ctx.half, ctx.comm = group, total_dev, residual is not None, half, comm
        _bump(running_mean, running_var)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, mean, invstd, mask = ctx.saved_tensors
        grad_out = grad_out.contiguous().to(x.dtype)
        n, c = x.shape
        lib = L.load()
        sfx = "_f16" if ctx.half else ""
        grad_x = torch.empty_like(x)
        grad_res = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[1]) else None
        gwb = torch.empty((2, c), dtype=torch.float32, device=x.device)          # this rank's grad_weight, grad_bias
        ws = L.workspace(lib.ts_bn_train_workspace_bytes(c), x.device)
        if ctx.group is None:
            L.check(getattr(lib, "ts_bn_act_train_backward" + sfx)(
                L.ptr(grad_out), L.ptr(mask), L.ptr(x), L.ptr(mean), L.ptr(invstd), L.ptr(weight), n, c, L.ptr(grad_x),
                L.ptr(grad_res), L.ptr(gwb[0]), L.ptr(gwb[1]), L.ptr(ws), ws.numel(), L.stream()),
                "ts_bn_act_train_backward" + sfx)
        elif ctx.comm is not None:
            sums = torch.empty((2, c), dtype=torch.float64, device=x.device)
            L.check(lib.ts_bn_sync_backward(ctx.comm, L.ptr(grad_out), L.ptr(mask), L.ptr(x), L.ptr(mean), L.ptr(invstd),
                                            L.ptr(weight), L.ptr(ctx.total_dev), n, c, 1 if ctx.half else 0, L.ptr(sums),
                                            L.ptr(grad_x), L.ptr(grad_res), L.ptr(gwb[0]), L.ptr(gwb[1]), L.ptr(ws),
                                            ws.numel(), L.stream()), "ts_bn_sync_backward")
        else:
            sums = torch.empty((2, c), dtype=torch.float64, device=x.device)
            L.check(getattr(lib, "ts_bn_sync_backward_reduce" + sfx)(
                L.ptr(grad_out), L.ptr(mask), L.ptr(x), L.ptr(mean), L.ptr(invstd), n, c, L.ptr(sums), L.ptr(gwb[0]),
                L.ptr(gwb[1]), L.ptr(ws), ws.numel(), L.stream()), "ts_bn_sync_backward_reduce" + sfx)
            c10d_sum(sums, ctx.group)
            if ctx.half:
                L.check(lib.ts_bn_act_backward_f16(L.ptr(grad_out), L.ptr(mask), L.ptr(x), L.ptr(mean), L.ptr(invstd),
                                                   L.ptr(weight), L.ptr(sums), L.ptr(ctx.total_dev), float(n), n, c,
                                                   L.ptr(grad_x), L.ptr(grad_res), L.ptr(ws), ws.numel(), L.stream()),
                        "ts_bn_act_backward_f16")
            else:
                L.check(lib.ts_bn_act_backward(L.ptr(grad_out), L.ptr(mask), L.ptr(x), L.ptr(mean), L.ptr(invstd),
                                               L.ptr(weight), L.ptr(sums), L.ptr(ctx.total_dev), float(n), n, c,
                                               L.ptr(grad_x), L.ptr(grad_res), L.stream()), "ts_bn_act_backward")
        return grad_x, grad_res, gwb[0], gwb[1], None, None, None, None, None, None, None


def batch_norm_act_train(x, weight, bias, running_mean, running_var, momentum, eps, relu=True, residual=None,
                         group=None, num_batches_tracked=None):
    """act(BN(x) [+ residual]) with batch statistics, fused elementwise passes; updates the running buffers and
    (when given) increments `num_batches_tracked`."""
    return _BatchNormActTrain.apply(x, residual, weight, bias, running_mean, running_var, num_batches_tracked,
                                    momentum, eps, relu, group)


def batch_norm_train(x, weight, bias, running_mean, running_var, momentum, eps, group=None, num_batches_tracked=None):
    """y = BN(x) with batch statistics (over all ranks of `group` when given); updates the running buffers."""
    return batch_norm_act_train(x, weight, bias, running_mean, running_var, momentum, eps, relu=False, residual=None,
                                group=group, num_batches_tracked=num_batches_tracked)
