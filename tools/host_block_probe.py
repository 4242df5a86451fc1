"""Diagnostic: host microseconds per fused conv block call (forward / backward) on a tiny problem (device work ~0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse import nn as spnn

torch.manual_seed(0)
rs = np.random.RandomState(0)
pc = np.unique(rs.randint(0, 12, size=(600, 3)), axis=0).astype(np.int32)
coords = torch.from_numpy(np.concatenate([pc, np.zeros((len(pc), 1), np.int32)], 1)).cuda()
conv = spnn.Conv3d(32, 32, 3).cuda()
bn = spnn.BatchNorm(32).cuda().train()
x = SparseTensor(torch.randn(len(pc), 32, device="cuda", requires_grad=True), coords, 1)
y = spnn.conv_bn_act(conv, bn, x)           # builds the kernel map
n = 300
from taseg_amd import _fast
for fused in (True, False, True, False, "python-node"):      # first two rounds warm the allocator and load the kernels
    if fused == "python-node":
        _fast._mod, fused = None, True
    spnn.modules._FUSED_BLOCK = fused
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        outs.append(spnn.conv_bn_act(conv, bn, x).F)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    loss = sum(o.sum() for o in outs)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"fused={fused}: forward {1e6 * (t1 - t0) / n:.1f} us/block, backward {1e6 * (t3 - t2) / n:.1f} us/block (incl. autograd engine)")


class Triv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a.view_as(a)

    @staticmethod
    def backward(ctx, g):
        return g


a = torch.randn(8, device="cuda", requires_grad=True)
t0 = time.perf_counter()
outs = [Triv.apply(a) for _ in range(n)]
t1 = time.perf_counter()
s = sum(o.sum() for o in outs)
t2 = time.perf_counter()
s.backward()
t3 = time.perf_counter()
print(f"trivial Function: apply {1e6 * (t1 - t0) / n:.1f} us, backward {1e6 * (t3 - t2) / n:.1f} us per node (incl. the sum nodes)")
t0 = time.perf_counter()
for _ in range(2000):
    torch.empty((100, 32), device="cuda")
t1 = time.perf_counter()
print(f"torch.empty {1e6 * (t1 - t0) / 2000:.2f} us")
