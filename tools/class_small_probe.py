import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import ts_oracle as O
from taseg_amd import backend as B
def cloud(seed, n, extent, batch=2):
    rs = np.random.RandomState(seed)
    c = np.unique(np.concatenate([rs.randint(0, extent, (n, 3)), rs.randint(0, batch, (n, 1))], 1), axis=0)
    return c[rs.permutation(len(c))].astype(np.int32)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for n, ext in ((900, 12), (2500, 16), (9000, 24)):
    c = cloud(n, n, ext)
    offs = O.get_kernel_offsets(3, 1, 1)
    km = B.build_kmap(T(c), T(c), T(offs))
    plan = B.conv_class_plan(km["nbr"])
    _, nbmaps, nbsizes = O.build_kmap(c, c, offs)
    total = len(nbmaps)
    for ci, co in ((256, 256), (384, 256), (128, 256), (64, 128), (192, 128), (128, 128), (96, 96), (32, 32)):
        rs = np.random.RandomState(1)
        x = rs.randn(len(c), ci).astype(np.float32); w = (rs.randn(27, ci, co) / np.sqrt(ci)).astype(np.float32); gy = rs.randn(len(c), co).astype(np.float32)
        y64 = O.conv_forward(x.astype(np.float64), w.astype(np.float64), nbmaps, nbsizes, (len(c), len(c)))
        gx64, _ = O.conv_backward(x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64), nbmaps, nbsizes)
        xt, wt, gt = T(x), T(w), T(gy)
        y = B.conv_gather_sum(B.conv_class_gemm(xt, wt, plan), plan["pos"], len(c))
        gx = B.conv_gather_sum(B.conv_class_gemm(gt, wt, plan, weight_transposed=True), plan["pos"], len(c))
        y2 = B.conv_gather_sum(B.conv_pair_gemm(xt, wt, km["nbmaps"], km["nboffs"], total, 0), km["pos_out"], len(c))
        gx2 = B.conv_gather_sum(B.conv_pair_gemm(gt, wt, km["nbmaps"], km["nboffs"], total, 1, weight_transposed=True), km["pos_in"], len(c))
        rel = lambda a, b: float(np.abs(a.double().cpu().numpy() - b).max()) / float(np.abs(b).max())
        print(f"n={len(c)} P={total} {ci}->{co}: class y {rel(y, y64):.1e} gx {rel(gx, gx64):.1e} | two-pass y {rel(y2, y64):.1e} gx {rel(gx2, gx64):.1e}")
