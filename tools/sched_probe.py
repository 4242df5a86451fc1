"""Experiment: split pair GEMM with its tiles launched in SPATIAL order (schedule table sorted by the output row of the
tile's first pair) vs the default offset-major order, on the real rulebooks of the bench batch.

    python tools/sched_probe.py [--iters 20]
"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)
lib = B.L.load()
lib.ts_debug_tile_sched.argtypes = [ctypes.c_void_p, ctypes.c_int32]
lib.ts_debug_tile_sched.restype = None


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


def schedule(km, col, bm=128):
    offs = km.nboffs.cpu().tolist()
    ks, p0s = [], []
    for k in range(len(offs) - 1):
        for p in range(offs[k], offs[k + 1], bm):
            ks.append(k)
            p0s.append(p)
    ks = torch.tensor(ks, dtype=torch.int32, device="cuda")
    p0s = torch.tensor(p0s, dtype=torch.int32, device="cuda")
    key = km.nbmaps_buf[p0s.long(), col].long() * 64 + ks.long()
    order = torch.argsort(key)
    return torch.stack([ks[order], p0s[order]], 1).contiguous()


print("layer                    | default fwd dgrad | spatial(out) fwd dgrad | spatial(in) fwd dgrad  (us)")
for s, ci, co in ((1, 96, 96), (1, 128, 96), (1, 32, 32), (2, 96, 96), (2, 32, 32), (2, 128, 96), (4, 128, 128), (4, 64, 64),
                  (8, 128, 128), (8, 256, 256), (16, 256, 256)):
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, ci, device="cuda")
    gy = torch.randn(n, co, device="cuda")
    w = torch.randn(27, ci, co, device="cuda") * 0.05
    row = f"s{s:<2d} {ci:3d}->{co:3d} P={P:8d} "
    ref = None
    for mode in ("default", 1, 0):
        if mode == "default":
            B.set_conv_impl(0)
        else:
            tab = schedule(km, mode)
            lib.ts_debug_tile_sched(tab.data_ptr(), tab.shape[0])
            B.set_conv_impl(9)
            B._conv_impl = 0
        f = lambda: B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)
        d = lambda: B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True)
        z = f()
        if ref is None:
            ref = z.clone()
        else:
            assert torch.equal(z, ref), "schedule changed the result"
        row += f"| {timed(f):7.1f} {timed(d):7.1f} "
    B.set_conv_impl(0)
    print(row, flush=True)
