"""A/B: split pair GEMM one tile per workgroup (impl 0) vs persistent cross-tile pipelined (impl 10), bench rulebooks."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

ap = argparse.ArgumentParser()
ap.add_argument("--impls", default="0,10")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
impls = [int(i) for i in args.impls.split(",")]
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
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


print("layer                    " + "".join(f"| impl {i}: fwd  dgrad (us) " for i in impls))
tot = {i: 0.0 for i in impls}
for s, ks, ci, co, reps in ((1, 3, 96, 96, 3), (1, 3, 128, 96, 1), (1, 3, 32, 32, 1), (2, 3, 96, 96, 3), (2, 3, 32, 32, 4), (2, 3, 128, 96, 1),
                           (4, 3, 128, 128, 3), (4, 3, 64, 64, 5), (4, 3, 192, 128, 1), (8, 3, 128, 128, 7), (8, 3, 256, 256, 3),
                           (8, 3, 384, 256, 1), (16, 3, 256, 256, 11), (1, 2, 32, 32, 1), (2, 2, 32, 32, 1), (4, 2, 64, 64, 1)):
    km = x.kmaps[((s, s, s), (ks,) * 3, (1, 1, 1) if ks == 3 else (2, 2, 2), (1, 1, 1))]
    n_in, n_out = km.sizes
    P = km.total
    xf = torch.randn(n_in, ci, device="cuda")
    gy = torch.randn(n_out, co, device="cuda")
    w = torch.randn(ks ** 3, ci, co, device="cuda") * 0.05
    row = f"s{s:<2d} k{ks} {ci:3d}->{co:3d} P={P:8d} "
    ref = None
    for impl in impls:
        B.set_conv_impl(impl)
        B._conv_impl = 0
        f = lambda: B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)
        d = lambda: B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True)
        zf, zd = f(), d()
        if ref is None:
            ref = (zf.clone(), zd.clone())
        else:
            assert torch.equal(zf, ref[0]) and torch.equal(zd, ref[1]), f"impl {impl} changed the result"
        tf, td = timed(f), timed(d)
        tot[impl] += reps * (tf + td)
        row += f"| {tf:7.1f} {td:7.1f} "
    B.set_conv_impl(0)
    print(row, flush=True)
print("weighted by the layer counts of mk34 (us per step, fwd + dgrad):", {i: round(v) for i, v in tot.items()})
