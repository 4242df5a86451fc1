"""Diagnostic: where the HOST time of the AMP training step goes (the step is host-bound at bs 2): wall time per phase of the
issue loop (enqueue only), the device time of the same steps, and a cProfile of 10 steps.   python tools/host_amp_probe.py [--fp32]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd.data.synthetic import make_model_cfg
from taseg_amd.data.stage import DevicePrefetcher
from taseg_amd.optim import FlatSGD
from taseg_amd.pcseg.model import build_network
from taseg_amd.torchsparse import SparseTensor

AMP = "--fp32" not in sys.argv
EARLY = "--early" in sys.argv
cfg = make_model_cfg("MinkUNet", in_dim=4, cr=1.0)
model = build_network(cfg, 20).cuda().train()
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, max_norm=10.0, amp=AMP)
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
offset = torch.tensor([len(coords)], device="cuda", dtype=torch.int32)
pf = DevicePrefetcher(lambda: {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset},
                      model.prepare, threaded=AMP)


def step(t=None):
    mark = (lambda: t.append(time.perf_counter())) if t is not None else (lambda: None)
    mark()
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=AMP):
        ret, _, _ = model(pf.next())
    mark()
    if EARLY:
        pf.prefetch_early()
    loss = ret["loss"].float().mean() * opt.loss_scale()
    mark()
    loss.backward()
    mark()
    opt.step()
    mark()
    pf.prefetch()
    mark()


for _ in range(5):
    step()
torch.cuda.synchronize()
rows = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
N = 20
for _ in range(N):
    t = []
    step(t)
    rows.append([1e3 * (b - a) for a, b in zip(t, t[1:])])
e1.record()
host = 1e3 * (time.perf_counter() - t0) / N
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t0) / N
import numpy as np
med = np.median(np.array(rows), 0)
print(f"{'AMP' if AMP else 'fp32'}: wall {wall:.2f} ms/step, host enqueue {host:.2f} ms/step, device span {e0.elapsed_time(e1) / N:.2f} ms/step")
print("host ms per phase (median): forward %.2f | loss %.2f | backward %.2f | optimizer %.2f | prefetch %.2f" % tuple(med))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
