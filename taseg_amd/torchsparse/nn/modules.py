"""`torchsparse.nn` modules (TS/torchsparse/nn/modules/{conv,norm,activation}.py)."""
import math

import torch
from torch import nn

from ..tensor import SparseTensor
from ..utils.misc import make_ntuple
from . import functional as F
from .utils import fapply

__all__ = ["Conv3d", "BatchNorm", "ReLU", "LeakyReLU"]


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
                        dilation=self.dilation, transposed=self.transposed)


class BatchNorm(nn.BatchNorm1d):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class ReLU(nn.ReLU):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)
