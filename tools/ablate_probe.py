"""Ablation of the split pair GEMM on the bench rulebooks: which part of the kernel holds it back?
impl 16 + bits: 1 = no Z stores, 2 = sequential rows instead of the gather, 4 = no MFMAs."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


modes = [(0, "full"), (17, "noZ"), (18, "seq"), (19, "seq+noZ"), (20, "noMMA"), (21, "noMMA+noZ"), (23, "noMMA+seq+noZ")]
print("layer                    " + " ".join(f"{n:>14s}" for _, n in modes))
for s, ci, co in ((1, 96, 96), (1, 32, 32), (2, 96, 96), (4, 128, 128), (8, 256, 256), (16, 256, 256)):
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, ci, device="cuda")
    w = torch.randn(27, ci, co, device="cuda") * 0.05
    row = f"s{s:<2d} {ci:3d}->{co:3d} P={P:8d} "
    for impl, _ in modes:
        B.set_conv_impl(impl)
        B._conv_impl = 0
        row += f"{timed(lambda: B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)):14.1f} "
    B.set_conv_impl(0)
    print(row, flush=True)
