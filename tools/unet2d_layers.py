"""Per-layer time of UNet2D's convolutions at the TIAF shape (10 frames of 384 x 1280, channels-last half, the shipped MIOpen
find-db): forward, data gradient, weight gradient of every nn.Conv2d as the vendor library runs them - the table that says which
layers are worth a hand-written kernel (profiles/r06_unet2d_layers.txt).

    python tools/unet2d_layers.py [frames]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.miopen_find_db()
from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import UNet2D  # noqa: E402


def timed(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    net = UNet2D(3, 20).cuda().half().to(memory_format=torch.channels_last)
    shapes = {}

    def hook(name):
        def f(mod, inp, out):
            shapes[name] = (tuple(inp[0].shape), tuple(out.shape))
        return f

    convs = {n: m for n, m in net.named_modules() if isinstance(m, torch.nn.Conv2d)}
    hs = [m.register_forward_hook(hook(n)) for n, m in convs.items()]
    x = torch.randn(frames, 3, 384, 1280, device="cuda").half().contiguous(memory_format=torch.channels_last)
    from taseg_amd.options import options
    with torch.no_grad(), options.override(image_conv_rows=False):
        s = net._encode(x)
        net.classifier(net._decode_u4(net._decode_u2(*s), s[1]))
    for h in hs:
        h.remove()
    del s
    torch.cuda.empty_cache()
    print("%-18s %-28s %5s %4s  %8s %8s %8s   %8s" % ("layer", "input -> output channels", "k", "dil", "fwd ms", "dgrad", "wgrad", "ideal*"))
    tot = [0.0, 0.0, 0.0]
    for name, conv in convs.items():
        (si, so) = shapes[name]
        xi = torch.randn(*si, device="cuda").half().contiguous(memory_format=torch.channels_last)
        gy = torch.randn(*so, device="cuda").half().contiguous(memory_format=torch.channels_last)
        w, b = conv.weight.detach(), conv.bias.detach()
        args = (conv.stride, conv.padding, conv.dilation, False, (0, 0), 1)
        f = timed(lambda: torch.ops.aten.convolution(xi, w, b, *args))
        d = timed(lambda: torch.ops.aten.convolution_backward(gy, xi, w, [b.numel()], *args, [True, False, False])) if si[1] > 3 else 0.0
        g = timed(lambda: torch.ops.aten.convolution_backward(gy, xi, w, [b.numel()], *args, [False, True, False]))
        ideal = 2.0 * (xi.numel() + gy.numel()) / 8e12 * 1e3      # both maps once at 8 TB/s
        tot = [tot[0] + f, tot[1] + d, tot[2] + g]
        print("%-18s %-28s %5d %4d  %8.3f %8.3f %8.3f   %8.3f" % (name, "%d @ %dx%d -> %d" % (si[1], si[2], si[3], so[1]), conv.kernel_size[0],
                                                                 conv.dilation[0], f, d, g, ideal))
        del xi, gy
    print("%-18s %-28s %5s %4s  %8.3f %8.3f %8.3f" % ("sum", "", "", "", *tot))
    print("* both maps once at 8 TB/s")


if __name__ == "__main__":
    main()
