"""sparse_quantize (TS/torchsparse/utils/quantize.py:9-46).

Host arrays (the reference's numpy data pipeline) are grouped with numpy exactly as the
reference does; ROCm tensors go to the HIP kernels (taseg_amd.backend.sparse_quantize).
Voxel order is ascending (x, y, z) ravel key; a voxel's representative is its first point.
"""
from itertools import repeat

import numpy as np
import torch

__all__ = ["sparse_quantize", "ravel_hash"]


def ravel_hash(x: np.ndarray) -> np.ndarray:
    """Row-major ravel of non-negative-shifted integer coordinates into one uint64 key."""
    assert x.ndim == 2, x.shape
    x = (x - x.min(axis=0)).astype(np.uint64, copy=False)
    extent = x.max(axis=0).astype(np.uint64) + np.uint64(1)
    key = np.zeros(x.shape[0], dtype=np.uint64)
    for d in range(x.shape[1] - 1):
        key = (key + x[:, d]) * extent[d + 1]
    return key + x[:, -1]


def _device_quantize(coords, voxel_size, return_index, return_inverse):
    from ... import backend as B
    vs = torch.as_tensor(voxel_size, dtype=torch.float32, device=coords.device)
    q = torch.floor(coords[:, :3].to(torch.float32) / vs).to(torch.int32)
    c4 = torch.cat([q, torch.zeros_like(q[:, :1])], dim=1).contiguous()
    index, inverse = B.sparse_quantize(c4)
    outs = [q[index.long()]]
    if return_index:
        outs.append(index.long())
    if return_inverse:
        outs.append(inverse.long())
    return outs[0] if len(outs) == 1 else outs


def sparse_quantize(coords, voxel_size=1, *, return_index: bool = False, return_inverse: bool = False):
    if isinstance(voxel_size, (float, int)):
        voxel_size = tuple(repeat(voxel_size, 3))
    assert isinstance(voxel_size, tuple) and len(voxel_size) == 3
    if isinstance(coords, torch.Tensor) and coords.is_cuda:
        return _device_quantize(coords, voxel_size, return_index, return_inverse)

    q = np.floor(np.asarray(coords) / np.array(voxel_size)).astype(np.int32)
    _, index, inverse = np.unique(ravel_hash(q), return_index=True, return_inverse=True)
    outs = [q[index]]
    if return_index:
        outs.append(index)
    if return_inverse:
        outs.append(inverse)
    return outs[0] if len(outs) == 1 else outs
