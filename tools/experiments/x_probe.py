"""Diagnostic (experiment): the three-product IEEE-half split pair GEMM (csrc/conv_pairs_x.hip, built apart into
csrc/build/libtaseg_x.so) next to the six-product bf16 split of the product path: time and error against float64.
    python tools/x_probe.py --stride 1 --cin 96 --cout 96"""
import argparse, ctypes, math, os, sys
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
ap.add_argument("--scale", type=float, default=1.0, help="magnitude of the gathered operand (gradients are ~1e-4)")
args = ap.parse_args()
X = ctypes.CDLL(os.path.join(os.path.dirname(L.__file__), "csrc", "build", "libtaseg_x.so"))
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)
s = args.stride
km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
n, P = km.sizes[0], km.total
torch.manual_seed(0)
xf = torch.randn(n, args.cin, device="cuda") * args.scale
xf[::7] *= 1e-3                                    # rows far below the tensor's largest magnitude
w = torch.randn(27, args.cin, args.cout, device="cuda") * 0.05
flops = 2.0 * P * args.cin * args.cout
print(f"stride {s}: {n} voxels, {P} pairs, {args.cin} -> {args.cout}, operand scale {args.scale}")


def pow2(t):
    return 2.0 ** (14 - math.floor(math.log2(float(t.abs().max()))))


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


def run_x(wt):
    z = torch.empty((P, args.cin if wt else args.cout), device="cuda")
    src = gy if wt else xf
    sx, sw = pow2(src), pow2(w)

    def go():
        rc = X.ts_debug_pair_gemm_x(ctypes.c_void_p(src.data_ptr()), src.shape[1], ctypes.c_void_p(w.data_ptr()), z.shape[1],
                                    ctypes.c_void_p(km.nbmaps_buf.data_ptr()), ctypes.c_void_p(km.nboffs.data_ptr()), 27,
                                    ctypes.c_int64(P), 1 if wt else 0, 1 if wt else 0, ctypes.c_float(sx), ctypes.c_float(sw),
                                    ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    return go, z


gy = torch.randn(n, args.cout, device="cuda") * args.scale
gy[::5] *= 1e-4
offs = km.nboffs.cpu().numpy()
maps = km.nbmaps_buf[:P].long()
kk = torch.repeat_interleave(torch.arange(27, device="cuda"), torch.tensor(offs[1:] - offs[:-1], device="cuda"))
sel = torch.randperm(P, device="cuda")[:200000]
for wt, name in ((0, "fwd"), (1, "dgrad")):
    src = gy if wt else xf
    rows = maps[sel, 1 if wt else 0]
    w64 = w.double().transpose(1, 2) if wt else w.double()
    ref = torch.bmm(src[rows].double().unsqueeze(1), w64[kk[sel]]).squeeze(1)
    scale_ref = float(ref.abs().max())
    zs = B.conv_pair_gemm(src, w, km.nbmaps_buf, km.nboffs, P, wt, weight_transposed=bool(wt))
    go, zx = run_x(wt)
    go()
    torch.cuda.synchronize()
    t6 = timed(lambda: B.conv_pair_gemm(src, w, km.nbmaps_buf, km.nboffs, P, wt, weight_transposed=bool(wt)))
    t3 = timed(go)
    rowscale = ref.abs().amax(1, keepdim=True).clamp_min(1e-30)
    for tag, z, t in (("bf16 x6", zs, t6), ("f16  x3", zx, t3)):
        d = (z[sel].double() - ref).abs()
        print(f"{name:6s} {tag}: {t:8.1f} us {flops / t / 1e6:7.1f} TF/s   max |err| / tensor max {float(d.max()) / scale_ref:.2e}   "
              f"max over rows of (row err / row max) {float((d.amax(1, keepdim=True) / rowscale).max()):.2e}")
