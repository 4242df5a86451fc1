"""Diagnostic (experiment): the three-product IEEE-half class GEMM with per-(row, slice) scales (tools/experiments/conv_class_x.hip,
libtaseg_exp.so) next to the six-product bf16 split of the product path: time and error against float64.
    python tools/experiments/class_x_probe.py --stride 1 --cin 96 --cout 96 [--scale 1e-4]"""
import argparse, ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
ap.add_argument("--scale", type=float, default=1.0, help="magnitude of the gathered operand (gradients are ~1e-4)")
args = ap.parse_args()
ctypes.CDLL(L.LIB_PATH, mode=ctypes.RTLD_GLOBAL)          # the experiments library refers to the product library's globals
X = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "libtaseg_exp.so"))
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)
s = args.stride
km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
n, P = km.sizes[0], km.total
nbr = km.nbr
torch.manual_seed(0)
xf = torch.randn(n, args.cin, device="cuda") * args.scale
xf[::7] *= 1e-3                                    # rows far below the tensor's largest magnitude
xf[:, ::5] *= 30.0                                 # columns far above the others inside a row
gy = torch.randn(n, args.cout, device="cuda") * args.scale
w = torch.randn(27, args.cin, args.cout, device="cuda") * 0.05
w[3] *= 1e-2                                       # an offset with small weights
plan = B.conv_class_plan(nbr)
eb = torch.tensor([14 - math.floor(math.log2(float(w[k].abs().max()))) for k in range(27)], dtype=torch.int32, device="cuda")
print(f"stride {s}: {n} voxels, {P} pairs, {args.cin} -> {args.cout}, operand scale {args.scale}")


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


def run_x(feat, wt):
    cols = args.cin if wt else args.cout
    zp = torch.empty((plan["m_pad"], cols), device="cuda")

    def go():
        rc = X.ts_debug_class_gemm_x(ctypes.c_void_p(feat.data_ptr()), feat.shape[1], ctypes.c_void_p(w.data_ptr()), 27, 3, cols,
                                     ctypes.c_void_p(plan["src"].data_ptr()), ctypes.c_int64(plan["m_pad"]),
                                     ctypes.c_void_p(plan["tile_info"].data_ptr()), ctypes.c_void_p(plan["n_tiles"].data_ptr()),
                                     1 if wt else 0, 1 if wt else 0, ctypes.c_void_p(eb.data_ptr()), ctypes.c_void_p(zp.data_ptr()),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    go()
    return go, zp


for wt, name in ((0, "fwd"), (1, "dgrad")):
    feat = gy if wt else xf
    cols = args.cin if wt else args.cout
    go_x, zx = run_x(feat, wt)
    go_s = lambda: B.conv_class_gemm(feat, w, plan, weight_transposed=bool(wt))
    zs = go_s()
    yx, ys = B.conv_gather_sum(zx, plan["pos"], n), B.conv_gather_sum(zs, plan["pos"], n)
    sel = torch.randperm(n, device="cuda")[:20000]
    ref = torch.zeros((len(sel), cols), dtype=torch.float64, device="cuda")
    for k in range(27):
        rows = nbr[k][sel].long()
        ok = rows >= 0
        wk = (w[26 - k].double().t() if wt else w[k].double())
        ref[ok] += feat[rows[ok]].double() @ wk
    scale = float(ref.abs().max())
    rowscale = ref.abs().amax(1).clamp_min(1e-300)
    e = [(float((y[sel].double() - ref).abs().max()) / scale, float(((y[sel].double() - ref).abs().amax(1) / rowscale).max())) for y in (ys, yx)]
    t_s, t_x = timed(go_s), timed(go_x)
    print(f"{name:6s} six bf16 products {t_s:7.1f} us   three f16 products, (row, slice) scales {t_x:7.1f} us   ratio {t_s / t_x:.2f}x   "
          f"max error / tensor max: {e[0][0]:.1e} vs {e[1][0]:.1e}   worst row error / row max: {e[0][1]:.1e} vs {e[1][1]:.1e}")
