"""Diagnostic: one conv layer's kernels (pair GEMM fwd / dgrad, gather-sum, weight gradient) on the real rulebook of a
synthetic 2 x 120k-point batch at a given stride.  Prints per-kernel time and TF/s; run under rocprofv3 --pmc
to read counters for exactly these kernels.

    python tools/conv_probe.py --stride 4 --cin 128 --cout 128 [--what fwd,dgrad,wgrad,gsum] [--iters 20]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

ap = argparse.ArgumentParser()
ap.add_argument("--stride", type=int, default=4)
ap.add_argument("--cin", type=int, default=128)
ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--what", default="fwd,dgrad,wgrad,gsum")
ap.add_argument("--impl", type=int, default=0)
ap.add_argument("--planes", action="store_true", help="hand the pre-split weight planes to the pair GEMM (what the training step does)")
args = ap.parse_args()

coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)
s = args.stride
km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
n = km.sizes[0]
P = km.total
print(f"stride {s}: {n} voxels, {P} pairs ({P / n:.1f} per voxel), Cin {args.cin} Cout {args.cout}")
B.set_conv_impl(args.impl)
xf = torch.randn(n, args.cin, device="cuda")
gy = torch.randn(n, args.cout, device="cuda")
w = torch.randn(27, args.cin, args.cout, device="cuda") * 0.05
flops = 2.0 * P * args.cin * args.cout


def timed(fn, label, fl=None, byts=None):
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
    extra = f"  {fl / us / 1e6:7.1f} TF/s" if fl else ""
    extra += f"  {byts / us / 1e3:7.1f} GB/s" if byts else ""
    print(f"{label:28s} {us:9.1f} us{extra}")


what = args.what.split(",")
if args.planes:
    from taseg_amd import _lib as L
    planes = torch.empty(3 * w.numel(), dtype=torch.int16, device="cuda")
    L.check(L.load().ts_conv_split_planes(w.data_ptr(), 27, args.cin, args.cout, planes.data_ptr(), L.stream()), "split")
    _pg = B.conv_pair_gemm

    def _hinted(*a, **k):
        L.load().ts_conv_planes_hint(w.data_ptr(), planes.data_ptr(), 27, args.cin, args.cout)
        return _pg(*a, **k)
    B.conv_pair_gemm = _hinted
if "fwd" in what:
    timed(lambda: B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0), "fwd", flops)
if "dgrad" in what:
    timed(lambda: B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True), "dgrad", flops)
if "wgrad" in what:
    timed(lambda: B.conv_wgrad(xf, gy, km.nbmaps_buf, km.nboffs, 27, col_a=0, max_pairs=P), "wgrad", flops)
if "shared" in what:
    # what sharing the gathered dY rows between the input gradient and the weight gradient could save at most: the same launches
    # with the dY column of the rulebook pointed at row 0 (those gathers come from cache), and with both columns at row 0
    m0 = km.nbmaps_buf.clone()
    m0[:, 1] = 0
    m00 = torch.zeros_like(km.nbmaps_buf)
    timed(lambda: B.conv_pair_gemm(gy, w, m0, km.nboffs, P, 1, weight_transposed=True), "dgrad, dY rows from cache", flops)
    timed(lambda: B.conv_wgrad(xf, gy, m0, km.nboffs, 27, col_a=0, max_pairs=P), "wgrad, dY rows from cache", flops)
    timed(lambda: B.conv_wgrad(xf, gy, m00, km.nboffs, 27, col_a=0, max_pairs=P), "wgrad, all rows from cache", flops)
if "f16" in what:
    w16, w16t = B.cast_weights_f16(w)
    xh, gyh = xf.half(), gy.half()
    timed(lambda: B.cast_weights_f16(w), "cast16")
    timed(lambda: B.conv_pair_gemm_f16(xh, w16t, km.nbmaps_buf, km.nboffs, P, 0), "fwd16", flops,
          P * (args.cin + args.cout) * 2)
    timed(lambda: B.conv_pair_gemm_f16(gyh, w16, km.nbmaps_buf, km.nboffs, P, 1), "dgrad16", flops,
          P * (args.cin + args.cout) * 2)
    timed(lambda: B.conv_wgrad_f16(xh, gyh, km.nbmaps_buf, km.nboffs, 27, col_a=0, max_pairs=P), "wgrad16", flops,
          P * (args.cin + args.cout) * 2)
    zh = B.conv_pair_gemm_f16(xh, w16t, km.nbmaps_buf, km.nboffs, P, 0)
    timed(lambda: B.conv_gather_sum_f16(zh, km.pos_out, n), "gsum16", None, P * args.cout * 2 + n * args.cout * 2 + 27 * n * 4)
if "gsum" in what:
    z = B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)
    timed(lambda: B.conv_gather_sum(z, km.pos_out, n), "gsum", None, P * args.cout * 4 + n * args.cout * 4 + 27 * n * 4)
