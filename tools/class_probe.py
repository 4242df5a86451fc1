"""Diagnostic (experiment): class-sorted implicit GEMM (csrc/conv_pairs_x.hip, ts_debug_class_gemm) + a 3-position pass 2
against pair GEMM + gather-sum on the bench rulebook.  The plan (rows sorted by the 9-bit neighbour mask of each offset group)
is built by ts_conv_class_plan.       python tools/class_probe.py --stride 1 --cin 96 --cout 96"""
import argparse, ctypes, os, sys
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
ap.add_argument("--half", action="store_true", help="IEEE-half rows (the autocast path)")
ap.add_argument("--ms", action="store_true", help="lexicographic stride-1 order (the multi-scan models' voxel order) instead of hash order")
ap.add_argument("--morton", type=int, default=0, help="stride-1 rows in Morton order of (x, y, z) >> MORTON-1 blocks (lexicographic inside a block)")
args = ap.parse_args()
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
if args.ms:
    import numpy as np
    c = coords.cpu().numpy()
    coords = torch.from_numpy(c[np.lexsort((c[:, 2], c[:, 1], c[:, 0], c[:, 3]))]).cuda()
if args.morton:
    import numpy as np
    c = coords.cpu().numpy().astype(np.int64)
    sh = args.morton - 1
    q = (c[:, :3] - c[:, :3].min(0)) >> sh
    code = np.zeros(len(c), dtype=np.int64)
    for b in range(16):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + (2 - a))
    order = np.lexsort((c[:, 2], c[:, 1], c[:, 0], code, c[:, 3]))
    coords = torch.from_numpy(c[order].astype(np.int32)).cuda()
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)
s = args.stride
km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
n, P = km.sizes[0], km.total
nbr = km.nbr                                             # [27, n]
torch.manual_seed(0)
xf = torch.randn(n, args.cin, device="cuda")
gy = torch.randn(n, args.cout, device="cuda")
w = torch.randn(27, args.cin, args.cout, device="cuda") * 0.05
flops = 2.0 * P * args.cin * args.cout

# ---- plan (device, csrc/conv_class.hip)
plan = B.conv_class_plan(nbr)
torch.cuda.synchronize()
pos, m_pad = plan["pos"], plan["m_pad"]
n_tiles = int(plan["n_tiles"][0])
info = plan["tile_info"][:n_tiles]
steps = int(sum(bin(int(v)).count("1") for v in info[:, 1].tolist()))
print(f"stride {s}: {n} voxels, {P} pairs, {args.cin} -> {args.cout}; Z' rows {n_tiles * 128} ({n_tiles * 128 / n:.2f} N), {n_tiles} tiles, "
      f"{steps} (tile, offset) steps ({steps / n_tiles:.2f} per tile; pair GEMM: {sum((int(v) + 127) // 128 for v in (km.nboffs[1:] - km.nboffs[:-1]).tolist())} tiles)")
t_plan = None


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


def class_gemm(feat, wt, c_out):
    holder = {}

    def go():
        holder["z"] = (B.conv_class_gemm_f16(feat, w, plan, weight_transposed=bool(wt)) if args.half else
                       B.conv_class_gemm(feat, w, plan, weight_transposed=bool(wt)))
    go()
    return go, holder["z"]


if args.half:
    xf, gy, w = xf.half(), gy.half(), w.half()
gsum = B.conv_gather_sum_f16 if args.half else B.conv_gather_sum


for wt, name, feat, c_out in ((0, "fwd", xf, args.cout), (1, "dgrad", gy, args.cin)):
    feat = gy if wt else xf
    if wt:
        two1 = ((lambda: B.conv_pair_gemm_f16(gy, w, km.nbmaps_buf, km.nboffs, P, 1)) if args.half else
                (lambda: B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True)))
        table = km.pos_in
    else:
        two1 = ((lambda: B.conv_pair_gemm_f16(xf, w, km.nbmaps_buf, km.nboffs, P, 0, natural=True)) if args.half else
                (lambda: B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)))
        table = km.pos_out
    z = two1()
    y2 = gsum(z, table, n)
    go, zp = class_gemm(feat, wt, c_out)
    go()
    y1 = gsum(zp, pos, n)
    err = float((y1.float() - y2.float()).abs().max()) / float(y2.float().abs().max())
    sel = torch.randperm(n, device="cuda")[:20000]
    ref = torch.zeros((len(sel), c_out), dtype=torch.float64, device="cuda")
    for k in range(27):
        src_rows = nbr[k][sel].long()
        ok = src_rows >= 0
        wk = (w[26 - k].double().t() if wt else w[k].double())
        ref[ok] += feat[src_rows[ok]].double() @ wk
    e64 = [float((y[sel].double() - ref).abs().max()) / float(ref.abs().max()) for y in (y2, y1)]
    t_a, t_b = timed(two1), timed(lambda: gsum(z, table, n))
    t_c, t_d = timed(go), timed(lambda: gsum(zp, pos, n))
    conv = B.conv_class_conv_f16 if args.half else B.conv_class_conv
    yf = conv(feat, w, plan, weight_transposed=bool(wt))
    t_f = timed(lambda: conv(feat, w, plan, weight_transposed=bool(wt)))
    same = bool(torch.equal(yf, y1))
    if t_plan is None:
        t_plan = timed(lambda: B.conv_class_plan(nbr))
        print(f"plan build {t_plan:.1f} us (once per batch and stride, staging stream)")
    print(f"{name:6s} two passes {t_a:7.1f} + {t_b:6.1f} = {t_a + t_b:7.1f} us   class-sorted {t_c:7.1f} + {t_d:6.1f} = {t_c + t_d:7.1f} us   "
          f"ratio {(t_a + t_b) / (t_c + t_d):.2f}x   finished in the product {t_f:7.1f} us ({(t_a + t_b) / t_f:.2f}x, same bits: {same})   "
          f"max |diff| / max |y| {err:.1e}   vs float64: two passes {e64[0]:.1e}, class-sorted {e64[1]:.1e}")
