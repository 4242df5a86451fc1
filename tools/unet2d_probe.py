"""UNet2D (the camera branch of TIAF) alone on the device: forward + backward of the stack `bench.py --workload tiaf` feeds it
(10 frames of 3 x 384 x 1280 at bs 2) in the four combinations of memory format (NCHW / channels_last) and arithmetic
(fp32 / torch.autocast fp16), each with the time its first pass took (MIOpen's solver search on a fresh box) and the steady
step.  One JSON record per mode on stdout.

    python tools/unet2d_probe.py [--frames 10] [--modes nchw_f32,nhwc_f32,nchw_amp,nhwc_amp] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--modes", default="nhwc_amp,nchw_amp,nhwc_f32,nchw_f32")
    ap.add_argument("--benchmark", type=int, default=-1, help="torch.backends.cudnn.benchmark (-1: leave the default)")
    args = ap.parse_args()
    if args.benchmark >= 0:
        torch.backends.cudnn.benchmark = bool(args.benchmark)
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import (ResBlock, ResContextBlock, UNet2D, UpBlock)  # noqa: F401
    torch.manual_seed(0)
    net = UNet2D(3, 20).cuda().train()
    x0 = torch.rand(args.frames, 3, args.height, args.width, device="cuda")

    def dense(net, x):
        # the dense part of UNet2D.forward (no gathers): what MIOpen / hipBLASLt see
        x0_ = net.stem(x)
        x1, s1 = net.stage1(x0_)
        x2, s2 = net.stage2(x1)
        x3, s3 = net.stage3(x2)
        x4, s4 = net.stage4(x3)
        x5 = net.mid_stage(x4)
        u1 = net.up1(x5, s4)
        u2 = net.up2(u1, s3)
        u3 = net.up3(u2, s2)
        u4 = net.up4(u3, s1)
        return net.classifier(u4), u4, u2

    for mode in args.modes.split(","):
        fmt, arith = mode.split("_")
        amp = arith == "amp"
        m = net.to(memory_format=torch.channels_last) if fmt == "nhwc" else net.to(memory_format=torch.contiguous_format)
        x = x0.contiguous(memory_format=torch.channels_last) if fmt == "nhwc" else x0.contiguous()

        def step():
            for p in m.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                logits, u4, u2 = dense(m, x)
            # gradients of the three maps the TIAF step takes gradients of
            loss = logits.float().mean() + u4.float().mean() + u2.float().mean()
            loss.backward()
            return logits

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = step()
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        step()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"mode": mode, "frames": args.frames, "first_pass_s": round(first, 2), "step_ms": round(1e3 * dt, 2),
                          "logits_dtype": str(out.dtype), "logits_channels_last": out.is_contiguous(memory_format=torch.channels_last),
                          "peak_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2),
                          "cudnn_benchmark": torch.backends.cudnn.benchmark}), flush=True)


if __name__ == "__main__":
    main()
