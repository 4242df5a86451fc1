"""Gradients of UNet2D's building blocks in torch.channels_last against the same module in the contiguous format, same inputs: which
library kernels of this PyTorch-ROCm can be trusted on channels-last maps?  (round 6: AvgPool2d's gradient cannot - profiles/r06_channels_last_op_check.txt)

    python tools/channels_last_op_check.py
"""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import ResContextBlock, ResBlock, UpBlock
torch.manual_seed(0)
CL = torch.channels_last
def check(name, make, shapes, dtype=torch.float32, train=True):
    mods = {}
    res = {}
    for fmt in ("nchw", "nhwc"):
        torch.manual_seed(1)
        m = make().cuda().to(dtype)
        m.train(train)
        for mm in m.modules():
            if isinstance(mm, torch.nn.Dropout2d): mm.eval()
        if fmt == "nhwc": m = m.to(memory_format=CL)
        torch.manual_seed(2)
        xs = [torch.randn(*s, device="cuda").to(dtype) for s in shapes]
        xs = [(x.contiguous(memory_format=CL) if fmt == "nhwc" else x).requires_grad_() for x in xs]
        out = m(*xs)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        torch.manual_seed(3)
        loss = sum((o.float() * torch.randn(o.shape, device="cuda")).sum() for o in outs)
        loss.backward()
        res[fmt] = [x.grad.float() for x in xs] + [p.grad.float() for p in m.parameters()]
        names = [f"x{i}" for i in range(len(xs))] + [n for n, _ in m.named_parameters()]
    worst = max(((float((a - b).norm() / a.norm().clamp_min(1e-20)), n) for a, b, n in zip(res["nchw"], res["nhwc"], names)))
    print(f"{name:28s} {str(dtype):14s} worst rel grad diff {worst[0]:.2e} ({worst[1]})")
for dt in (torch.float32, torch.float16):
    check("Conv2d 3x3 + bias", lambda: torch.nn.Conv2d(32, 32, 3, padding=1), [(4, 32, 32, 64)], dt)
    check("Conv2d 3x3 dil2", lambda: torch.nn.Conv2d(32, 32, 3, padding=2, dilation=2), [(4, 32, 32, 64)], dt)
    check("Conv2d 1x1", lambda: torch.nn.Conv2d(3, 32, 1), [(4, 3, 32, 64)], dt)
    check("BatchNorm2d train", lambda: torch.nn.BatchNorm2d(32), [(4, 32, 32, 64)], dt if dt == torch.float32 else torch.float32)
    check("LeakyReLU", lambda: torch.nn.LeakyReLU(), [(4, 32, 32, 64)], dt)
    check("AvgPool2d", lambda: torch.nn.AvgPool2d(3, stride=2, padding=1), [(4, 32, 32, 64)], dt)
    check("PixelShuffle", lambda: torch.nn.PixelShuffle(2), [(4, 64, 16, 32)], dt)
    check("ResContextBlock", lambda: ResContextBlock(32, 32), [(4, 32, 32, 64)], dt if dt == torch.float32 else torch.float32)
    check("ResBlock pool", lambda: ResBlock(32, 64, 0.2, pooling=True), [(4, 32, 32, 64)], dt if dt == torch.float32 else torch.float32)
    check("UpBlock", lambda: UpBlock(256, 256, 0.2, mid_filters=256 // 4 + 256), [(4, 256, 4, 8), (4, 256, 8, 16)], dt if dt == torch.float32 else torch.float32)
# BN with half input (autocast-like: fp32 params, half activations)
for fmt in ("nchw", "nhwc"):
    torch.manual_seed(1)
    bn = torch.nn.BatchNorm2d(32).cuda().train()
    x = torch.randn(4, 32, 32, 64, device="cuda").half()
    if fmt == "nhwc": x = x.contiguous(memory_format=CL); bn = bn.to(memory_format=CL)
    x.requires_grad_()
    torch.manual_seed(3)
    y = bn(x); (y.float() * torch.randn(y.shape, device="cuda")).sum().backward()
    print("BN half input", fmt, y.dtype, float(x.grad.float().norm()), float(bn.weight.grad.norm()), float(bn.bias.grad.norm()))
