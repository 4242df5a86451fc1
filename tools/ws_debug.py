"""Diagnostic: parameters after 3 steps with and without the weight gradient on a second stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd import _fast
from taseg_amd.optim import FlatSGD
from taseg_amd.data.synthetic import make_model_cfg
from taseg_amd.pcseg.model import build_network
from taseg_amd.torchsparse import SparseTensor
DEV = "cuda"
coords, feats, labels, _ = bench.make_scans(3, 2, 20000, "minkunet")
offset = torch.tensor([len(coords)], device=DEV, dtype=torch.int32)


def run(side, amp, steps=3):
    torch.manual_seed(0)
    cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5)
    model = build_network(cfg, 20).to(DEV).train()
    opt = FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=1e-4, max_norm=10.0, amp=amp)
    _fast.wgrad_stream(side)
    losses, grads = [], None
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            ret, _, _ = model({"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset})
        (ret["loss"].float().mean() * opt.loss_scale()).backward()
        if i == 0:
            torch.cuda.synchronize()
            grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        opt.step()
        losses.append(float(ret["loss"].detach()))
    torch.cuda.synchronize()
    _fast.wgrad_stream(False)
    return losses, {n: p.detach().clone() for n, p in model.named_parameters()}, grads


for amp in (False, True):
    a = run(False, amp)
    b = run(False, amp)
    c = run(True, amp)
    print("amp", amp, "losses", a[0], b[0], c[0])
    for tag, other in (("repeat without", b), ("with side stream", c)):
        bad_g = [n for n in a[2] if not torch.equal(a[2][n], other[2][n])]
        bad_p = [n for n in a[1] if not torch.equal(a[1][n], other[1][n])]
        print(f"  {tag}: first-step gradients differing {len(bad_g)} / {len(a[2])}, parameters after 3 steps differing {len(bad_p)} / {len(a[1])}")
        for n in bad_g[:6]:
            print("     grad", n, float((a[2][n] - other[2][n]).abs().max()), float(a[2][n].abs().max()))
        for n in bad_p[:6]:
            print("     param", n, float((a[1][n] - other[1][n]).abs().max()))
