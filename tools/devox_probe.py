"""Diagnostic: devoxelize backward (float atomics per run) at the three strides the U-Net devoxelises at."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.pcseg.model.segmentor.voxel.minkunet.minkunet import MinkUNetBackbone

coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
plan = MinkUNetBackbone._index_plan(coords, coords.float())


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


n = coords.shape[0]
for s, c in ((1, 96), (4, 128), (16, 256)):
    key = (s, s, s)
    idx, w = plan["tri_idx"][key], plan["tri_w"][key]
    order = plan["tri_order"].get(key)
    m = plan["cmaps"][key].shape[0]
    g = torch.randn(n, c, device="cuda")
    live = int(((idx >= 0) & (w != 0)).sum())
    runs_order = order if not isinstance(order, tuple) else (B.devox_order(idx, m) if s > 1 else None)
    t = timed(lambda: B.devoxelize_backward_runs(g, idx, w, m, runs_order))
    csr = order if isinstance(order, tuple) else B.devox_csr(idx, w, m)
    tc = timed(lambda: B.devoxelize_backward_csr(g, w, csr, m))
    tb = timed(lambda: B.devox_csr(idx, w, m))
    tf = timed(lambda: B.devoxelize_forward_cuda(torch.randn(m, c, device="cuda"), idx, w))
    print(f"stride {s}: {n} points -> {m} voxels, C={c}, live (point, corner) pairs {live} ({live / n:.2f} per point): "
          f"backward runs/atomics {t:.1f} us, inverse-map gather {tc:.1f} us (map build {tb:.1f} us), forward {tf:.1f} us (incl. randn)")
