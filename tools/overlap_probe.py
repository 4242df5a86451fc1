"""Diagnostic: what would co-scheduling the input-gradient pair GEMM and the weight gradient of one layer buy?
Times dgrad and wgrad back to back on one stream against the same two launches on two streams."""
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
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
it = 30
for s, ci, co in ((16, 256, 256), (8, 128, 128), (8, 256, 256), (8, 384, 256), (4, 128, 128), (4, 64, 64), (2, 96, 96), (1, 32, 32), (1, 96, 96)):
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, ci, device="cuda")
    gy = torch.randn(n, co, device="cuda")
    w = torch.randn(27, ci, co, device="cuda") * 0.05
    dgrad = lambda: B.conv_gather_sum(B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True), km.pos_in, n)
    wgrad = lambda: B.conv_wgrad(xf, gy, km.nbmaps_buf, km.nboffs, 27, col_a=0, max_pairs=P)

    def seq():
        dgrad(); wgrad()

    def par():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        main.wait_stream(side)

    res = []
    for fn in (seq, par):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / it * 1e3)
    print(f"s{s:<2d} {ci:3d}->{co:3d} P={P:8d}: dgrad+gsum then wgrad {res[0]:7.1f} us   two streams {res[1]:7.1f} us", flush=True)

# ---- a CHAIN of layers (backward order of the decoder / encoder tail): per layer the side stream waits for one event of the main
# stream (the point where the layer's output gradient exists) and runs the weight gradient; ONE join at the end of the chain
chain = [(1, 96, 96)] * 3 + [(2, 96, 96)] * 3 + [(4, 128, 128)] * 3 + [(8, 256, 256)] * 3 + [(16, 256, 256)] * 6 + [(8, 128, 128)] * 5 + \
        [(4, 64, 64)] * 5 + [(2, 32, 32)] * 3
ops = []
for s, ci, co in chain:
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, ci, device="cuda")
    gy = torch.randn(n, co, device="cuda")
    w = torch.randn(27, ci, co, device="cuda") * 0.05
    ops.append((lambda gy=gy, w=w, km=km, P=P, n=n: B.conv_gather_sum(B.conv_pair_gemm(gy, w, km.nbmaps_buf, km.nboffs, P, 1, weight_transposed=True), km.pos_in, n),
                lambda xf=xf, gy=gy, km=km, P=P: B.conv_wgrad(xf, gy, km.nbmaps_buf, km.nboffs, 27, col_a=0, max_pairs=P)))


def chain_seq():
    for dg, wg in ops:
        dg(); wg()


def chain_par():
    for dg, wg in ops:
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            wg()
        dg()
    main.wait_stream(side)


for name, fn in (("one stream", chain_seq), ("weight gradients on a side stream, one join", chain_par), ("one stream", chain_seq),
                 ("weight gradients on a side stream, one join", chain_par)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"chain of {len(ops)} layers, {name}: {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
