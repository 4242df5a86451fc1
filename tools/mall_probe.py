"""Diagnostic: does the Infinity Cache (256 MB) hold Z between the pair GEMM and the gather-sum?  Times gather-sum right
after the pair GEMM that wrote its Z ("hot") and after 1.5 GB of unrelated writes in between ("cold"), and a chunked
form in which a stride's rulebook is processed in row ranges whose Z fits the cache.

    python tools/mall_probe.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)
flush = torch.empty(384 * 1024 * 1024, dtype=torch.float32, device="cuda")      # 1.5 GB


def ev():
    return torch.cuda.Event(enable_timing=True)


for s, c in ((4, 128), (2, 96), (1, 96), (1, 32)):
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, c, device="cuda")
    w = torch.randn(27, c, c, device="cuda") * 0.05
    res = {}
    for mode in ("hot", "cold"):
        ts = []
        for it in range(6):
            z = B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)
            if mode == "cold":
                flush.zero_()
            e0, e1 = ev(), ev()
            e0.record()
            y = B.conv_gather_sum(z, km.pos_out, n)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[mode] = sorted(ts[1:])[len(ts[1:]) // 2]
    zmb = P * c * 4 / 1e6
    byts = P * c * 4 + n * c * 4 + 27 * n * 4
    print(f"stride {s} C {c}: Z {zmb:.0f} MB; gather-sum hot {res['hot']:.1f} us ({byts / res['hot'] / 1e6:.2f} TB/s), "
          f"cold {res['cold']:.1f} us ({byts / res['cold'] / 1e6:.2f} TB/s)")
