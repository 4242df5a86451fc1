"""SparseTensor / PointTensor containers (API of TS/torchsparse/tensor.py:10-105).

Field names, the ``F``/``C``/``s`` aliases, the shared ``cmaps``/``kmaps`` dictionaries
and the PointTensor caches are the contract the pcseg model code relies on
(minkunet.py:386-394, minkunet/utils.py:31-107); the implementation is ours.
"""
from typing import Any, Dict, Tuple

import torch

from .utils.misc import make_ntuple

__all__ = ["SparseTensor", "PointTensor"]


def _alias(field):
    return property(lambda self: getattr(self, field), lambda self, v: setattr(self, field, v))


class _Movable:
    """cpu()/cuda()/to()/detach() act in place on the listed tensor fields and return self."""
    _tensor_fields: Tuple[str, ...] = ()

    def _apply(self, fn):
        for name in self._tensor_fields:
            setattr(self, name, fn(getattr(self, name)))
        return self

    def cpu(self):
        return self._apply(lambda t: t.cpu())

    def cuda(self):
        return self._apply(lambda t: t.cuda())

    def detach(self):
        return self._apply(lambda t: t.detach())

    def to(self, device, non_blocking: bool = True):
        return self._apply(lambda t: t.to(device, non_blocking=non_blocking))


class SparseTensor(_Movable):
    """feats [N, C] + coords [N, 4] int32 (x, y, z, batch) at a tensor stride.

    ``cmaps`` {stride -> coords} and ``kmaps`` {(stride, kernel, stride, dilation) -> kernel map}
    are shared by reference between every tensor derived from the same input.
    """
    _tensor_fields = ("coords", "feats")

    def __init__(self, feats: torch.Tensor, coords: torch.Tensor, stride=1) -> None:
        self.feats = feats
        self.coords = coords
        self.stride = make_ntuple(stride, ndim=3)
        self.cmaps: Dict[Tuple[int, ...], torch.Tensor] = {}
        self.kmaps: Dict[Tuple[Any, ...], Any] = {}

    F = _alias("feats")
    C = _alias("coords")

    @property
    def s(self):
        return self.stride

    @s.setter
    def s(self, stride):
        self.stride = make_ntuple(stride, ndim=3)

    def _like(self, feats):
        out = SparseTensor(feats, self.coords, self.stride)
        out.cmaps, out.kmaps = self.cmaps, self.kmaps
        return out

    def __add__(self, other):
        return self._like(self.feats + other.feats)


class PointTensor(_Movable):
    """Per-point features F [N, C] and float coordinates C [N, 4]; caches the trilinear
    maps per voxel stride (``idx_query`` / ``weights``) and the voxelisation maps
    (``additional_features['idx_query' | 'counts']``)."""
    _tensor_fields = ("F", "C")

    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = coords
        self.idx_query = {} if idx_query is None else idx_query
        self.weights = {} if weights is None else weights
        self.additional_features = {"idx_query": {}, "counts": {}}

    def __add__(self, other):
        out = PointTensor(self.F + other.F, self.C, self.idx_query, self.weights)
        out.additional_features = self.additional_features
        return out
