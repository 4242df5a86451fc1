"""torchsparse.cat (TS/torchsparse/operators.py:10-17): channel concat of co-located tensors."""
from typing import List

import torch

from .tensor import SparseTensor

__all__ = ["cat"]


def cat(inputs: List[SparseTensor]) -> SparseTensor:
    head = inputs[0]
    return head._like(torch.cat([t.feats for t in inputs], dim=1))
