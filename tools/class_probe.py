"""Diagnostic (experiment): class-sorted implicit GEMM (csrc/conv_pairs_x.hip, ts_debug_class_gemm) + a 3-position pass 2
against pair GEMM + gather-sum on the bench rulebook.  The plan (rows sorted by the 9-bit neighbour mask of each offset group)
is built here with torch ops.       python tools/class_probe.py --stride 1 --cin 96 --cout 96"""
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
args = ap.parse_args()
X = ctypes.CDLL(os.path.join(os.path.dirname(L.__file__), "csrc", "build", "libtaseg_x.so"))
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
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

# ---- plan
srcs, infos, pos = [], [], torch.full((3, n), -1, dtype=torch.int32, device="cuda")
base = 0
sh = torch.arange(9, device="cuda")
for g in range(3):
    sub = nbr[9 * g:9 * g + 9]                           # offsets 9 g + kl: one z-plane
    bits = ((sub >= 0).long() << sh[:, None]).sum(0)
    live = bits.nonzero().squeeze(1)
    order = live[torch.sort(bits[live], stable=True).indices]
    ns = order.numel()
    npad = (ns + 127) // 128 * 128
    src = torch.full((9, npad), -1, dtype=torch.int32, device="cuda")
    src[:, :ns] = sub[:, order]
    bp = torch.zeros(npad, dtype=torch.long, device="cuda")
    bp[:ns] = bits[order]
    tm = torch.zeros(npad // 128, dtype=torch.long, device="cuda")
    for b in range(9):
        tm |= (((bp.view(-1, 128) >> b) & 1).amax(1)) << b
    infos.append(torch.stack([g + 4 * (base // 128 + torch.arange(npad // 128, device="cuda")), tm], 1).int())
    pos[g, order] = (base + torch.arange(ns, device="cuda")).int()
    srcs.append(src)
    base += npad
src = torch.cat(srcs, 1).contiguous()                    # [9, m_pad]
info = torch.cat(infos, 0).contiguous()
pop = sum(((info[:, 1] >> b) & 1) for b in range(9))
info = info[torch.sort(pop, descending=True, stable=True).indices].contiguous()      # longest tiles first
m_pad, n_tiles = src.shape[1], info.shape[0]
steps = int(sum(bin(int(v)).count("1") for v in info[:, 1].tolist()))
print(f"stride {s}: {n} voxels, {P} pairs, {args.cin} -> {args.cout}; Z' rows {m_pad} ({m_pad / n:.2f} N), {n_tiles} tiles, "
      f"{steps} (tile, offset) steps ({steps / n_tiles:.2f} per tile; pair GEMM: {sum((int(v) + 127) // 128 for v in (km.nboffs[1:] - km.nboffs[:-1]).tolist())} tiles)")


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
    zp = torch.empty((m_pad, c_out), device="cuda")

    def go():
        rc = X.ts_debug_class_gemm(ctypes.c_void_p(feat.data_ptr()), feat.shape[1], ctypes.c_void_p(w.data_ptr()), c_out,
                                   ctypes.c_void_p(src.data_ptr()), ctypes.c_int64(m_pad), ctypes.c_void_p(info.data_ptr()),
                                   n_tiles, 27, wt, ctypes.c_void_p(zp.data_ptr()),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    return go, zp


for wt, name, feat, c_out in ((0, "fwd", xf, args.cout), (1, "dgrad", gy, args.cin)):
    if wt:
        two1 = lambda: B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True)
        table = km.pos_in
    else:
        two1 = lambda: B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)
        table = km.pos_out
    z = two1()
    y2 = B.conv_gather_sum(z, table, n)
    go, zp = class_gemm(feat, wt, c_out)
    go()
    y1 = B.conv_gather_sum(zp, pos, n)
    err = float((y1 - y2).abs().max()) / float(y2.abs().max())
    t_a, t_b = timed(two1), timed(lambda: B.conv_gather_sum(z, table, n))
    t_c, t_d = timed(go), timed(lambda: B.conv_gather_sum(zp, pos, n))
    print(f"{name:6s} two passes {t_a:7.1f} + {t_b:6.1f} = {t_a + t_b:7.1f} us   class-sorted {t_c:7.1f} + {t_d:6.1f} = {t_c + t_d:7.1f} us   "
          f"ratio {(t_a + t_b) / (t_c + t_d):.2f}x   max |diff| / max |y| {err:.1e}")
