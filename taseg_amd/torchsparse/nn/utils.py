"""get_kernel_offsets / fapply (TS/torchsparse/nn/utils/{kernel,apply}.py)."""
import itertools
from typing import Callable

import torch

from ..tensor import SparseTensor
from ..utils.misc import make_ntuple

__all__ = ["get_kernel_offsets", "fapply"]

_offset_cache = {}


def get_kernel_offsets(size, stride=1, dilation=1, device="cpu") -> torch.Tensor:
    """[K, 3] int32 offsets in the order that indexes `Conv3d.kernel`'s first axis
    (kernel.py:11-32): per axis arange(-size//2 + 1, size//2 + 1) * stride * dilation;
    odd volumes enumerate z outermost / x innermost (MinkowskiEngine weight layout),
    even volumes x outermost / z innermost."""
    size, stride, dilation = (make_ntuple(v, ndim=3) for v in (size, stride, dilation))
    key = (size, stride, dilation, str(device))
    hit = _offset_cache.get(key)
    if hit is not None:
        return hit
    axes = [[(v * stride[a] * dilation[a]) for v in range(-size[a] // 2 + 1, size[a] // 2 + 1)] for a in range(3)]
    if (size[0] * size[1] * size[2]) % 2 == 1:
        rows = [(x, y, z) for z, y, x in itertools.product(axes[2], axes[1], axes[0])]
    else:
        rows = list(itertools.product(axes[0], axes[1], axes[2]))
    out = torch.tensor(rows, dtype=torch.int, device=device)
    _offset_cache[key] = out
    return out


def fapply(input: SparseTensor, fn: Callable[..., torch.Tensor], *args, **kwargs) -> SparseTensor:
    """Apply a dense row-wise function to the features, keep coordinates and caches."""
    return input._like(fn(input.feats, *args, **kwargs))
