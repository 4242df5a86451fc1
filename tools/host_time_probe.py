"""Diagnostic: host enqueue time vs device time of one training step (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd.data.synthetic import make_model_cfg
from taseg_amd.pcseg.model import build_network
from taseg_amd.torchsparse import SparseTensor

cfg = make_model_cfg("MinkUNet", in_dim=4, cr=1.0)
model = build_network(cfg, 20).cuda().train()
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
offset = torch.tensor([len(coords)], device="cuda", dtype=torch.int32)

from taseg_amd.data.stage import DevicePrefetcher
def make_batch():
    return {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset}
pf = None if "--no-prefetch" in sys.argv else DevicePrefetcher(make_batch, model.prepare)

def step():
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True)
    ret, _, _ = model(make_batch() if pf is None else pf.next())
    t.append(time.perf_counter())
    ret["loss"].mean().backward()
    t.append(time.perf_counter())
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()
    t.append(time.perf_counter())
    if pf is not None:
        pf.prefetch()
    t.append(time.perf_counter())
    return t

for _ in range(4):
    step()
torch.cuda.synchronize()
rows = []
t_all = time.perf_counter()
for _ in range(10):
    rows.append(step())
host_done = time.perf_counter()
torch.cuda.synchronize()
dev_done = time.perf_counter()
import numpy as np
r = np.array(rows)
d = np.diff(r, axis=1).mean(0) * 1e3
print("host ms/step: forward %.1f  backward %.1f  clip+opt %.1f  stage next batch %.1f   total %.1f" % (d[0], d[1], d[2], d[3], d.sum()))
print("10 steps: host finished after %.1f ms, device after %.1f ms" % ((host_done - t_all) * 1e3, (dev_done - t_all) * 1e3))
