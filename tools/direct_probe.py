"""Diagnostic: the one-pass (direct class plan) 2x2x2 strided / transposed convolutions next to pair GEMM + pass 2 on the bench
rulebooks, per map and direction (csrc/conv_class.hip).    python tools/direct_probe.py [--amp]"""
import argparse
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from taseg_amd import backend as B  # noqa: E402
from taseg_amd.data.synthetic import make_model_cfg  # noqa: E402
from taseg_amd.pcseg.model import build_network  # noqa: E402
from taseg_amd.torchsparse import SparseTensor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--amp", action="store_true")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
coords, feats, labels, npts = bench.make_scans(0, 2, 120000, "minkunet")
model = build_network(make_model_cfg("MinkUNet", in_dim=4, cr=0.125, num_layer=[1] * 8), 20).cuda().train()
plan = model.prepare({"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords),
                      "offset": torch.tensor([len(coords)], device="cuda", dtype=torch.int32)})


def timed(fn, reps=args.reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


half = args.amp
dt = torch.float16 if half else torch.float32
cg = B.conv_class_gemm_f16 if half else B.conv_class_gemm
pg = B.conv_pair_gemm_f16 if half else B.conv_pair_gemm
gs = B.conv_gather_sum_f16 if half else B.conv_gather_sum
# mk34: down convs 32->32 (s1), 32->32 (s2), 64->64 (s4), 128->128 (s8); up convs 256->256 (s16->8), 256->128, 128->96, 96->96
layers = {1: [(32, 32, False), (96, 96, True)], 2: [(32, 32, False), (128, 96, True)], 4: [(64, 64, False), (256, 128, True)],
          8: [(128, 128, False), (256, 256, True)]}
print(f"{'map':>10s} {'conv':>22s} {'dir':>6s} {'two-pass us':>12s} {'direct us':>10s} {'ratio':>6s}")
for s, convs in layers.items():
    km = plan["kmaps"][((s,) * 3, (2, 2, 2), (2, 2, 2), (1, 1, 1))]
    d = km.build_direct_plans()
    nf, nc = km.sizes
    for ci, co, transposed in convs:
        w = torch.randn(8, ci, co, device="cuda").to(dt)
        nat = dict(natural=True) if half else {}
        wtr = {} if half else dict(weight_transposed=True)
        if not transposed:      # strided conv: fwd fine -> coarse (down plan), dgrad coarse -> fine (up plan)
            x, g = torch.randn(nf, ci, device="cuda").to(dt), torch.randn(nc, co, device="cuda").to(dt)
            cases = [("fwd", lambda: gs(pg(x, w, km.nbmaps_buf, km.nboffs, km.total, 0, **nat), km.pos_out, nc), lambda: cg(x, w, d["down"])),
                     ("dgrad", lambda: gs(pg(g, w, km.nbmaps_buf, km.nboffs, km.total, 1, **wtr), km.pos_in, nf),
                      lambda: cg(g, w, d["up"], weight_transposed=True))]
        else:                   # transposed conv: fwd coarse -> fine (up plan), dgrad fine -> coarse (down plan)
            x, g = torch.randn(nc, ci, device="cuda").to(dt), torch.randn(nf, co, device="cuda").to(dt)
            cases = [("fwd", lambda: gs(pg(x, w, km.nbmaps_buf, km.nboffs, km.total, 1, **nat), km.pos_in, nf), lambda: cg(x, w, d["up"])),
                     ("dgrad", lambda: gs(pg(g, w, km.nbmaps_buf, km.nboffs, km.total, 0, **wtr), km.pos_out, nc),
                      lambda: cg(g, w, d["down"], weight_transposed=True))]
        for name, two, direct in cases:
            a, b = timed(two), timed(direct)
            print(f"{f's{s} {nf}->{nc}':>10s} {f'{ci}->{co}' + (' transposed' if transposed else ''):>22s} {name:>6s} {a:12.1f} {b:10.1f} {a / b:6.2f}")
