"""Diagnostic: per (kernel, shape) time table of one training step of the bench workload (HIP events per launch)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import backend as B
from taseg_amd.data.synthetic import make_model_cfg
from taseg_amd.pcseg.model import build_network
from taseg_amd.torchsparse import SparseTensor

cfg = make_model_cfg("MinkUNet", in_dim=4, cr=1.0)
model = build_network(cfg, 20).cuda().train()
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
offset = torch.tensor([len(coords)], device="cuda", dtype=torch.int32)


def step():
    model.zero_grad(set_to_none=True)
    ret, _, _ = model({"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset})
    ret["loss"].mean().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
B.profile_begin()
N = 3
for _ in range(N):
    step()
torch.cuda.synchronize()
rec = B.profile_end()
tab = collections.defaultdict(lambda: [0, 0.0])
cache = {}
for kind, e0, e1, m in rec:
    if kind == "conv_wgrad" and "nboffs" in m:
        key = m["nboffs"].data_ptr()
        if key not in cache:
            cache[key] = int(m["nboffs"][-1])
        p = cache[key]
    else:
        p = m.get("pairs", 0)
    k = (m["name"], p, m["c_red"], m["c_out"], m["k"])
    tab[k][0] += 1
    tab[k][1] += e0.elapsed_time(e1)
rows = sorted(tab.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in tab.values()) / N
print(f"conv kernels total {tot:.2f} ms/step")
for (name, p, cr, co, k), (n, ms) in rows[:60]:
    fl = 2.0 * p * cr * co if cr else 0
    us = ms / n * 1e3
    print(f"{ms / N:6.3f} ms/step  x{n / N:4.1f}  {us:7.1f} us  {fl / us / 1e6 if fl else 0:6.1f} TF/s  P={p:8d} K={k:2d} {cr:4d}->{co:4d}  {name}")
