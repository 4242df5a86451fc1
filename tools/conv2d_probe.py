"""ts_conv3x3c32_rows (csrc/conv2d_rows.hip) against torch's Conv2d on channels-last half stacks: values (forward, data gradient) and
time per call at the TIAF shape (10 x 32 x 384 x 1280), plain and dilated.

    python tools/conv2d_probe.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taseg_amd import backend as B  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    torch.manual_seed(0)
    if "general" in sys.argv[1:]:
        return general()
    for shape, dil in (((2, 32, 19, 45), 1), ((2, 32, 19, 45), 2), ((10, 32, 384, 1280), 1), ((10, 32, 384, 1280), 2)):
        conv = torch.nn.Conv2d(32, 32, 3, padding=dil, dilation=dil).cuda().half().to(memory_format=torch.channels_last)
        x = torch.randn(*shape, device="cuda").half().contiguous(memory_format=torch.channels_last).requires_grad_()
        want = conv(x)
        gy = torch.randn_like(want)
        (gx_want,) = torch.autograd.grad(want, x, gy)
        p0, p1 = B.conv3x3c32_pack(conv.weight.detach(), 0), B.conv3x3c32_pack(conv.weight.detach(), 1)
        got = B.conv3x3c32_rows(x.detach(), p0, conv.bias.detach().float(), dil)
        gx = B.conv3x3c32_rows(gy, p1, None, dil)
        ref = torch.nn.functional.conv2d(x.detach().double(), conv.weight.detach().double(), conv.bias.detach().double(), padding=dil, dilation=dil) \
            if shape[0] != 10 else want.double()
        gw = B.conv3x3c32_wgrad(x.detach(), gy, conv.weight.detach(), dil)
        _, gw_want, _ = torch.ops.aten.convolution_backward(gy, x.detach(), conv.weight.detach(), None, [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1,
                                                            [False, True, False])
        if shape[0] == 10:       # (the large stack: against the vendor library's half result; float64 only on the small ones)
            gw_ref = gw_want.double()
        else:
            w64 = conv.weight.detach().double().requires_grad_()
            (gw_ref,) = torch.autograd.grad(torch.nn.functional.conv2d(x.detach().double(), w64, None, padding=dil, dilation=dil), w64, gy.double())
        rec = {"shape": shape, "dilation": dil,
               "wgrad_rel_err_vs_f64": float((gw.double() - gw_ref).norm() / gw_ref.norm()),
               "torch_wgrad_rel_err_vs_f64": float((gw_want.double() - gw_ref).norm() / gw_ref.norm()),
               "fwd_max_err_vs_f64": float((got.double() - ref).abs().max()), "torch_fwd_max_err_vs_f64": float((want.double() - ref).abs().max()),
               "dgrad_max_abs_diff_vs_torch": float((gx.float() - gx_want.float()).abs().max()), "dgrad_scale": float(gx_want.float().abs().max())}
        if shape[0] == 10:
            xd = x.detach()
            rec["ours_fwd_us"] = timed(lambda: B.conv3x3c32_rows(xd, p0, conv.bias.detach().float(), dil))
            rec["torch_fwd_us"] = timed(lambda: conv(xd))
            rec["ours_dgrad_us"] = timed(lambda: B.conv3x3c32_rows(gy, p1, None, dil))
            rec["torch_dgrad_us"] = timed(lambda: torch.ops.aten.convolution_backward(gy, xd, conv.weight, None, [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1, [True, False, False]))
            rec["torch_wgrad_bias_us"] = timed(lambda: torch.ops.aten.convolution_backward(gy, xd, conv.weight, [32], [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1, [False, True, True]))
            rec["ours_wgrad_us"] = timed(lambda: B.conv3x3c32_wgrad(xd, gy, conv.weight.detach(), dil))
            rec["torch_wgrad_us"] = timed(lambda: torch.ops.aten.convolution_backward(gy, xd, conv.weight, None, [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1, [False, True, False]))
            byts = 2 * xd.numel() * 2
            rec["ours_fwd_hbm_frac"] = byts / rec["ours_fwd_us"] / 1e3 / 8000.0
        print(json.dumps(rec), flush=True)
    general()


def general():
    """the general kernel (C_in <= 96) at the decoder's shapes: time against the vendor library, values against it"""
    import bench
    bench.miopen_find_db()
    for name, (t, ci, h, w), co in (("up4.conv1", (10, 56, 384, 1280), 96), ("up3.conv1", (10, 96, 192, 640), 96),
                                    ("stage2.conv2", (10, 32, 192, 640), 64)):
        conv = torch.nn.Conv2d(ci, co, 3, padding=1).cuda().half().to(memory_format=torch.channels_last)
        x = torch.randn(t, ci, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)
        gy = torch.randn(t, co, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)
        wd, bd = conv.weight.detach(), conv.bias.detach()
        p0, p1 = B.conv3x3_rows_pack(wd, 0), B.conv3x3_rows_pack(wd, 1)
        args = ([1, 1], [1, 1], [1, 1], False, [0, 0], 1)
        want = conv(x)
        got = B.conv3x3_rows(x, p0, bd.float(), co)
        gx_want = torch.ops.aten.convolution_backward(gy, x, wd, None, *args, [True, False, False])[0]
        gx = B.conv3x3_rows(gy, p1, None, ci)
        rec = {"layer": name, "shape": [t, ci, h, w], "c_out": co,
               "fwd_max_abs_diff_vs_torch": float((got.float() - want.float()).abs().max()), "fwd_scale": float(want.float().abs().max()),
               "dgrad_max_abs_diff_vs_torch": float((gx.float() - gx_want.float()).abs().max()), "dgrad_scale": float(gx_want.float().abs().max()),
               "ours_fwd_us": timed(lambda: B.conv3x3_rows(x, p0, bd.float(), co)), "torch_fwd_us": timed(lambda: conv(x)),
               "ours_dgrad_us": timed(lambda: B.conv3x3_rows(gy, p1, None, ci)),
               "torch_dgrad_us": timed(lambda: torch.ops.aten.convolution_backward(gy, x, wd, None, *args, [True, False, False])),
               "torch_wgrad_us": timed(lambda: torch.ops.aten.convolution_backward(gy, x, wd, None, *args, [False, True, False])),
               "ours_wgrad_bias_us": timed(lambda: B.conv3x3_wgrad(x, gy, wd, True)),
               "torch_bias_sum_us": timed(lambda: gy.sum(dim=(0, 2, 3), dtype=torch.float32))}
        gw, gb = B.conv3x3_wgrad(x, gy, wd, True)
        gw_want = torch.ops.aten.convolution_backward(gy, x, wd, None, *args, [False, True, False])[1]
        rec["wgrad_rel_diff_vs_torch"] = float((gw.float() - gw_want.float()).norm() / gw_want.float().norm())
        rec["bias_rel_diff_vs_torch"] = float((gb - gy.sum(dim=(0, 2, 3), dtype=torch.float32)).norm() / gb.norm())
        rec["ours_fwd_hbm_frac"] = 2 * (x.numel() + gy.numel()) / rec["ours_fwd_us"] / 1e3 / 8000.0
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
