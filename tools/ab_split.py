"""A/B of the split-bf16 (impl 0) and f32-MFMA (impl 5) full-tile kernels on the MinkUNet layer shapes, one process.

    python tools/ab_split.py [--impls 0,5] [--iters 20]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

ap = argparse.ArgumentParser()
ap.add_argument("--impls", default="0,5")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=2)
args = ap.parse_args()
impls = [int(i) for i in args.impls.split(",")]
coords, feats, labels, _ = bench.make_scans(0, args.batch, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


print("layer               " + "".join(f"| impl {i}: fwd  dgrad  wgrad (us) " for i in impls))
for s, ci, co in ((16, 256, 256), (8, 128, 128), (8, 256, 256), (8, 384, 256), (4, 128, 128), (4, 64, 64), (4, 256, 128),
                  (2, 96, 96), (2, 32, 64), (2, 192, 96), (1, 32, 32), (1, 96, 96), (1, 128, 96)):
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, ci, device="cuda")
    gy = torch.randn(n, co, device="cuda")
    w = torch.randn(27, ci, co, device="cuda") * 0.05
    row = f"s{s:<2d} {ci:3d}->{co:3d} P={P:8d} "
    for impl in impls:
        B.set_conv_impl(impl)
        t = [timed(lambda: B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)),
             timed(lambda: B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True)),
             timed(lambda: B.conv_wgrad(xf, gy, km.nbmaps_buf, km.nboffs, 27, col_a=0, max_pairs=P))]
        row += "| " + " ".join(f"{v:7.1f}" for v in t) + f"  ({2e-6 * P * ci * co / t[0]:5.1f} TF/s fwd) "
    B.set_conv_impl(0)
    print(row, flush=True)
