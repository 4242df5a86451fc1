"""Diagnostic: cProfile of the host side of the training step."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd.data.synthetic import make_model_cfg
from taseg_amd.pcseg.model import build_network
from taseg_amd.torchsparse import SparseTensor

cfg = make_model_cfg("MinkUNet", in_dim=4, cr=1.0)
model = build_network(cfg, 20).cuda().train()
from taseg_amd.optim import FlatSGD
AMP = "--amp" in sys.argv
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, max_norm=10.0, amp=AMP)
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
offset = torch.tensor([len(coords)], device="cuda", dtype=torch.int32)

from taseg_amd.data.stage import DevicePrefetcher
pf = DevicePrefetcher(lambda: {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords),
                               "offset": offset}, model.prepare)

def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=AMP):
        ret, _, _ = model(pf.next())
    (ret["loss"].float().mean() * opt.loss_scale()).backward()
    opt.step()
    pf.prefetch()

for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:12000])
