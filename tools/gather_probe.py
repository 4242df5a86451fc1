"""Diagnostic: pass 2 of the two-pass convolution (gather-sum) on the layer shapes of the benchmark, fp32 and half rows.
Run it twice - as is (live positions compacted in LDS, gather_list_kernel) and with TASEG_GATHER_POSITIONS=1 (K position
registers per lane, gather_sum_kernel) - and compare; the sums are bit-identical (tests/test_gpu_ops.py).
profiles/r02_v9_gather_forms_probe.txt also holds the numbers of a ROW-MAJOR Z (products of an output row stored next to
each other through a slot table: pass 2 15-40 % faster, the pair GEMM's scattered row stores 5-30 % slower - not adopted).

    python tools/gather_probe.py
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


def timed(fn, reps=10):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts[1:])[len(ts[1:]) // 2]


form = "K position registers" if os.environ.get("TASEG_GATHER_POSITIONS") else "LDS lists"
for s, c in ((1, 96), (2, 96), (4, 128), (8, 256), (16, 256), (1, 32), (2, 64)):
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, c, device="cuda")
    w = torch.randn(27, c, c, device="cuda") * 0.05
    z = B.conv_pair_gemm(xf, w, km.nbmaps_buf, km.nboffs, P, 0)
    t32 = timed(lambda: B.conv_gather_sum(z, km.pos_out, n))
    zh = z.half()
    t16 = timed(lambda: B.conv_gather_sum_f16(zh, km.pos_out, n))
    b32 = P * c * 4 + n * c * 4 + 27 * n * 4
    print(f"{form}: stride {s} C {c} ({P / n:.1f} pairs/row): fp32 {t32:.1f} us ({b32 / t32 / 1e6:.2f} TB/s), half {t16:.1f} us")
