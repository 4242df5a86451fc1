"""make_ntuple (TS/torchsparse/utils/utils.py:9-19)."""
import torch

__all__ = ["make_ntuple"]


def make_ntuple(x, ndim: int):
    if isinstance(x, torch.Tensor):
        x = [int(v) for v in x.reshape(-1).tolist()]
    if isinstance(x, int):
        return (x,) * ndim
    x = tuple(x)
    assert len(x) == ndim, x
    return x
