"""sparse_collate / sparse_collate_fn (TS/torchsparse/utils/collate.py:11-59):
stack per-scan SparseTensors into one batch by appending the batch index as 4th coordinate."""
from typing import Any, List

import numpy as np
import torch

__all__ = ["sparse_collate", "sparse_collate_fn"]


def _as_tensor(a):
    return torch.tensor(a) if isinstance(a, np.ndarray) else a


def sparse_collate(inputs: List[Any]):
    from ..tensor import SparseTensor
    stride = inputs[0].stride
    all_coords, all_feats = [], []
    for b, item in enumerate(inputs):
        item.coords, item.feats = _as_tensor(item.coords), _as_tensor(item.feats)
        assert isinstance(item.coords, torch.Tensor), type(item.coords)
        assert isinstance(item.feats, torch.Tensor), type(item.feats)
        assert item.stride == stride, (item.stride, stride)
        bcol = torch.full((item.coords.shape[0], 1), b, dtype=torch.int, device=item.coords.device)
        all_coords.append(torch.cat((item.coords, bcol), dim=1))
        all_feats.append(item.feats)
    return SparseTensor(coords=torch.cat(all_coords, dim=0), feats=torch.cat(all_feats, dim=0), stride=stride)


def sparse_collate_fn(inputs: List[Any]) -> Any:
    from ..tensor import SparseTensor
    if not isinstance(inputs[0], dict):
        return inputs
    batch = {}
    for key, first in inputs[0].items():
        column = [sample[key] for sample in inputs]
        if isinstance(first, dict):
            batch[key] = sparse_collate_fn(column)
        elif isinstance(first, np.ndarray):
            batch[key] = torch.stack([torch.tensor(v) for v in column], dim=0)
        elif isinstance(first, torch.Tensor):
            batch[key] = torch.stack(column, dim=0)
        elif isinstance(first, SparseTensor):
            batch[key] = sparse_collate(column)
        else:
            batch[key] = column
    return batch
