"""Diagnostic: output-stationary single-launch kernel (ts_conv_nbr) against the two-pass path on the small-channel layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for s, ci, co in ((1, 32, 32), (2, 32, 64), (2, 64, 64), (4, 64, 128), (1, 96, 96), (1, 4, 32)):
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, ci, device="cuda")
    w = torch.randn(27, ci, co, device="cuda") * 0.05
    two = timed(lambda: B.conv_gather_sum(B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0), km.pos_out, n))
    one = timed(lambda: B.conv_nbr(xf, w, km.nbr, n))
    print(f"s{s} {ci}->{co} N={n} P={P}: two-pass {two:7.1f} us   conv_nbr {one:7.1f} us", flush=True)
