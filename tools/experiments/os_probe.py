"""Diagnostic: the output-stationary convolution kernel (csrc/conv_os.hip) next to pair GEMM + gather-sum on the real
rulebook of the synthetic 2 x 120k-point batch.    python tools/os_probe.py --stride 1 --cin 96 --cout 96"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B, _lib as L
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

ap = argparse.ArgumentParser()
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--cin", type=int, default=96)
ap.add_argument("--cout", type=int, default=96)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--ablate", action="store_true")
ap.add_argument("--ms", action="store_true", help="lexicographic stride-1 order (the multi-scan models' voxel order)")
args = ap.parse_args()
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
if args.ms:
    c = coords.cpu().numpy()
    import numpy as np
    coords = torch.from_numpy(c[np.lexsort((c[:, 2], c[:, 1], c[:, 0], c[:, 3]))]).cuda()
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)
s = args.stride
km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
n, P = km.sizes[0], km.total
xf = torch.randn(n, args.cin, device="cuda")
gy = torch.randn(n, args.cout, device="cuda")
w = torch.randn(27, args.cin, args.cout, device="cuda") * 0.05
pl = torch.empty(3 * w.numel(), dtype=torch.int16, device="cuda")
L.check(L.load().ts_conv_split_planes(w.data_ptr(), 27, args.cin, args.cout, pl.data_ptr(), L.stream()), "split")
flops = 2.0 * P * args.cin * args.cout
print(f"stride {s}: {n} voxels, {P} pairs ({P / n:.1f}/voxel), {args.cin} -> {args.cout}")


def timed(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    print(f"{label:28s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF/s")
    return us


def two_pass_fwd():
    z = B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)
    return B.conv_gather_sum(z, km.pos_out, n)


def two_pass_dgrad():
    z = B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True)
    return B.conv_gather_sum(z, km.pos_in, n)


a = timed(two_pass_fwd, "fwd   pair GEMM + gather-sum")
b = timed(lambda: B.conv_os(xf, pl, w.shape, km.nbr), "fwd   output-stationary")
c = timed(two_pass_dgrad, "dgrad pair GEMM + gather-sum")
d = timed(lambda: B.conv_os(gy, pl, w.shape, km.nbr, weight_transposed=True), "dgrad output-stationary")
print(f"ratio fwd {a / b:.2f}x  dgrad {c / d:.2f}x   equal bits: fwd {torch.equal(two_pass_fwd(), B.conv_os(xf, pl, w.shape, km.nbr))} "
      f"dgrad {torch.equal(two_pass_dgrad(), B.conv_os(gy, pl, w.shape, km.nbr, weight_transposed=True))}")

if args.ablate:
    for bits, what in ((1, "no W staging"), (2, "no MFMA"), (4, "no tile update"), (8, "no gathered rows"), (6, "no MFMA, no tile update"),
                       (7, "no W, MFMA, update"), (15, "nothing but barriers + lists")):
        L.load().ts_debug_conv_os(bits)
        timed(lambda: B.conv_os(xf, pl, w.shape, km.nbr), f"fwd OS ablate {bits:2d} {what}")
    L.load().ts_debug_conv_os(0)
