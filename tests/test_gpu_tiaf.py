"""TIAF on the GPU: the image -> point gather kernel against plain indexing,
and `MinkUNetMsMm` (UNet2D + UNet3D + fused-cloud MinkUNet + fusion head, five losses) against the logits / losses /
gradients the REAL reference produced for the same inputs and parameters (tests/golden/model_minkunet_ms_mm.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from taseg_amd.data.synthetic import TIAF_CFG, fill_parameters, make_model_cfg  # noqa: E402

LOGIT_TOL = 1e-3


def _reference_gather(feat, pix, pbatch, frame_end, shift):
    """oracle.ts_oracle.image_gather with torch indexing (differentiable, for the adjoint check)."""
    outs, start = [], 0
    for b, end in enumerate(frame_end.tolist()):
        tall = feat[start:end].permute(0, 2, 3, 1).reshape(-1, feat.shape[3], feat.shape[1])
        p = pix[pbatch == b].long()
        outs.append(tall[p[:, 0] >> shift, p[:, 1] >> shift])
        start = end
    return torch.cat(outs, 0)


def _in_layout(t, layout):
    """the same values in channels-last ("nhwc") or plain contiguous ("nchw") memory"""
    return t.contiguous(memory_format=torch.channels_last) if layout == "nhwc" else t.contiguous()


@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
@pytest.mark.parametrize("shift,c", [(0, 96), (2, 128), (0, 1), (0, 20), (0, 3)])
def test_image_gather_matches_indexing(shift, c, layout):
    from taseg_amd import backend as B
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import image_gather
    g = torch.Generator().manual_seed(5)
    frames, H, W = [3, 2, 4], 24, 40
    frame_end = torch.tensor(np.cumsum(frames), dtype=torch.int32, device="cuda")
    feat = _in_layout(torch.randn(sum(frames), c, H >> shift, W >> shift, generator=g).cuda(), layout).requires_grad_()
    n = 5000
    pbatch = torch.sort(torch.randint(0, 3, (n,), generator=g)).values.int().cuda()     # batch-sorted, as collated
    rows = torch.stack([torch.randint(0, frames[b] * H, (1,), generator=g) for b in pbatch.tolist()]).view(-1)
    pix = torch.stack([rows, torch.randint(0, W, (n,), generator=g)], 1).float().cuda()
    out, err = image_gather(feat, pix, pbatch, frame_end, H, W, shift)
    want = _reference_gather(feat, pix, pbatch, frame_end, shift)
    assert int(err) == 0 and torch.equal(out, want)
    from oracle import ts_oracle as O
    assert np.array_equal(out.detach().cpu().numpy(), O.image_gather(feat.detach().cpu().numpy(), pix.cpu().numpy(),
                                                                     pbatch.cpu().numpy(), frame_end.cpu().numpy(), shift))
    w = torch.randn(n, c, generator=g).cuda()
    (g_ours,) = torch.autograd.grad((out * w).sum(), feat)
    (g_ref,) = torch.autograd.grad((want * w).sum(), feat)
    assert float((g_ours - g_ref).abs().max()) <= 1e-5 * max(1.0, float(g_ref.abs().max()))
    # a pixel beyond the sample's frames is reported, not read
    bad = pix.clone()
    bad[0, 0] = frames[0] * H + 1
    plan = B.image_plan(bad, pbatch, frame_end, sum(frames), H, W, shift)
    assert int(plan["err"]) == 1
    got = B.image_gather_rows_forward(feat.detach(), plan) if layout == "nhwc" else B.image_gather_forward(feat.detach(), plan)
    assert torch.equal(got[1:], want.detach()[1:]) and not got[0].any()       # the other rows are what they were, the bad one zeros


@pytest.mark.parametrize("dtype", [torch.float16, torch.uint8, torch.int64])
def test_image_row_gather_moves_any_dtype(dtype):
    """the channels-last gather is a pure move: half maps (autocast), byte images, integer label maps come out bit for bit"""
    from taseg_amd import backend as B
    g = torch.Generator().manual_seed(6)
    frames, H, W, c = [2, 3], 16, 24, 5
    T = sum(frames)
    frame_end = torch.tensor(np.cumsum(frames), dtype=torch.int32, device="cuda")
    base = torch.randint(0, 200, (T, c, H, W), generator=g)
    feat = base.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    n = 3000
    pbatch = torch.sort(torch.randint(0, 2, (n,), generator=g)).values.int().cuda()
    rows = torch.stack([torch.randint(0, frames[b] * H, (1,), generator=g) for b in pbatch.tolist()]).view(-1)
    pix = torch.stack([rows, torch.randint(0, W, (n,), generator=g)], 1).float().cuda()
    plan = B.image_plan(pix, pbatch, frame_end, T, H, W, 0)
    out = B.image_gather_rows_forward(feat, plan)
    want = _reference_gather(feat, pix, pbatch, frame_end, 0)
    assert out.dtype == dtype and torch.equal(out, want)
    with pytest.raises(ValueError, match="channels_last"):
        B.image_gather_rows_forward(feat.contiguous(), plan)


@pytest.mark.parametrize("layout,dtype", [("nhwc", torch.float32), ("nhwc", torch.float16), ("nchw", torch.float32)])
@pytest.mark.parametrize("shift,c", [(0, 96), (2, 128), (0, 20)])
def test_image_gather_at_tiaf_size_through_path(shift, c, layout, dtype):
    """>= 20k FOV points on a camera-sized stack (minkunet_mk34_cr10_fsa_tiaf.yaml: 384 x 1280), channels-last rows (fp32 and the
    fp16 maps of an autocast step) and NCHW planes: the raster-order gather against plain indexing and the oracle, and the adjoint
    ADDED into the map's other gradient in place (unet2d._image_gather_through) against autograd's own sum of the two gradients -
    run-to-run identical (no atomics)."""
    from oracle import ts_oracle as O
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import _image_gather_through as image_gather_through, image_plan
    g = torch.Generator().manual_seed(11)
    frames, H, W = [2, 1], 384, 1280
    T = sum(frames)
    frame_end = torch.tensor(np.cumsum(frames), dtype=torch.int32, device="cuda")
    feat = _in_layout(torch.randn(T, c, H >> shift, W >> shift, generator=g).cuda().to(dtype), layout).requires_grad_()
    n = 60000
    pbatch = torch.sort(torch.randint(0, 2, (n,), generator=g)).values.int().cuda()
    # scan-line-like pixels: a few rows, neighbouring columns, several points per pixel
    rows = torch.stack([torch.randint(0, frames[b] * H, (1,), generator=g) for b in pbatch.tolist()]).view(-1)
    rows = (rows // 6) * 6
    pix = torch.stack([rows, torch.randint(0, W // 3, (n,), generator=g) * 3], 1).float().cuda()
    plan = image_plan(pix, pbatch, frame_end, T, H, W, shift)
    assert int(plan["err"]) == 0
    through, out = image_gather_through(feat, plan)
    want = _reference_gather(feat, pix, pbatch, frame_end, shift)
    assert out.dtype == dtype and torch.equal(out, want)
    assert np.array_equal(out.detach().float().cpu().numpy(), O.image_gather(feat.detach().float().cpu().numpy(), pix.cpu().numpy(),
                                                                             pbatch.cpu().numpy(), frame_end.cpu().numpy(), shift))
    w = torch.randn(n, c, generator=g).cuda().to(dtype)
    dense = _in_layout(torch.randn(feat.shape, generator=g).cuda().to(dtype), layout)
    loss = (out * w).sum() + (through * dense).sum()             # the map's second consumer
    (g_ours,) = torch.autograd.grad(loss, feat, retain_graph=True)
    (g_again,) = torch.autograd.grad(loss, feat)
    assert torch.equal(g_ours, g_again)                          # deterministic
    assert g_ours.dtype == dtype and g_ours.is_contiguous(memory_format=torch.channels_last if layout == "nhwc" else torch.contiguous_format)
    # against a float64 evaluation of the same sum (fp16 maps: one rounding of the fp32 run sum into the map's gradient, where
    # index_put(accumulate) in half - what the reference does under autocast - rounds after every point)
    f64 = feat.detach().double().requires_grad_()
    (g_ref,) = torch.autograd.grad((_reference_gather(f64, pix, pbatch, frame_end, shift) * w.double()).sum() + (f64 * dense.double()).sum(), f64)
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert float((g_ours.double() - g_ref).abs().max()) <= tol * max(1.0, float(g_ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(3, 32, 24, 40), (2, 64, 17, 33), (1, 96, 5, 7), (2, 20, 8, 6)])
def test_channels_last_average_pool_equals_the_contiguous_module(shape, dtype):
    """UNet2D's AvgPool2d(3, stride 2, padding 1) on channels-last maps runs on the library's kernels (csrc/image.hip): forward and
    gradient against torch's own module in the CONTIGUOUS format (whose gradient is right; the channels-last gradient kernel of this
    PyTorch-ROCm is not - that is why the kernels exist), odd sizes included; deterministic"""
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import _pool
    g = torch.Generator().manual_seed(4)
    pool = torch.nn.AvgPool2d(kernel_size=(3, 3), stride=2, padding=1)
    x = torch.randn(*shape, generator=g).cuda().to(dtype)
    w = torch.randn(shape[0], shape[1], (shape[2] - 1) // 2 + 1, (shape[3] - 1) // 2 + 1, generator=g).cuda().to(dtype)
    a = x.clone().requires_grad_()
    want = pool(a)
    (ga,) = torch.autograd.grad((want.float() * w.float()).sum(), a)
    b = x.contiguous(memory_format=torch.channels_last).requires_grad_()
    got = _pool(pool, b)
    assert got.shape == want.shape and got.dtype == dtype and got.is_contiguous(memory_format=torch.channels_last)
    (gb,) = torch.autograd.grad((got.float() * w.float()).sum(), b, retain_graph=True)
    (gb2,) = torch.autograd.grad((got.float() * w.float()).sum(), b)
    tol = 1e-6 if dtype == torch.float32 else 2e-3
    assert float((got.float() - want.float()).abs().max()) <= tol * max(1.0, float(want.float().abs().max()))
    assert float((gb.float() - ga.float()).abs().max()) <= tol * max(1.0, float(ga.float().abs().max()))
    assert torch.equal(gb, gb2) and gb.is_contiguous(memory_format=torch.channels_last)
    # a contiguous input keeps the module's own path
    assert torch.equal(_pool(pool, x), pool(x))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(3, 32, 24, 40), (2, 96, 17, 33), (1, 256, 5, 7), (4, 64, 48, 160)])
def test_fused_leaky_relu_batch_norm_equals_the_two_modules(shape, dtype):
    """UNet2D's `bn(act(conv(x)))` tail on a channels-last map as one node (csrc/bn.hip ts_leaky_bn_train_*) against nn.LeakyReLU +
    nn.BatchNorm2d in training mode evaluated in float64: output, the three gradients, running statistics and the batch counter;
    run-to-run identical"""
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import _act_bn
    g = torch.Generator().manual_seed(7)
    c = shape[1]
    x = (torch.randn(*shape, generator=g) * 1.5 + 0.3).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(*shape, generator=g).cuda()
    act = torch.nn.LeakyReLU()

    def make_bn(dt):
        bn = torch.nn.BatchNorm2d(c).cuda().to(dt).train()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g))
            bn.running_mean.copy_(torch.randn(c, generator=g))
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        return bn
    g.manual_seed(8)
    ours = make_bn(torch.float32)
    g.manual_seed(8)
    ref = make_bn(torch.float64)
    a = x.clone().requires_grad_()
    out = _act_bn(act, ours, a)
    assert out.dtype == dtype and out.is_contiguous(memory_format=torch.channels_last) and out.grad_fn.__class__.__name__.startswith("_LeakyBatchNormRows")
    ga, gw, gb = torch.autograd.grad((out.float() * wt).sum(), (a, ours.weight, ours.bias), retain_graph=True)
    ga2, gw2, gb2 = torch.autograd.grad((out.float() * wt).sum(), (a, ours.weight, ours.bias))
    assert torch.equal(ga, ga2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)
    b = x.double().clone().requires_grad_()
    want = ref(act(b))
    gb_x, gb_w, gb_b = torch.autograd.grad((want * wt.double()).sum(), (b, ref.weight, ref.bias))
    tol = 2e-5 if dtype == torch.float32 else 4e-3
    rel = lambda p, q: float((p.double() - q).norm() / q.norm().clamp_min(1e-30))      # noqa: E731
    assert float((out.double() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    assert rel(ga, gb_x) <= tol and rel(gw, gb_w) <= tol and rel(gb, gb_b) <= tol, (rel(ga, gb_x), rel(gw, gb_w), rel(gb, gb_b))
    assert int(ours.num_batches_tracked) == 1
    assert torch.allclose(ours.running_mean.double(), ref.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(ours.running_var.double(), ref.running_var, rtol=1e-3 if dtype == torch.float16 else 1e-4, atol=1e-5)
    # the block's residual sum in the same pass (unet2d.py:31,64): bit-equal to adding it to the node's own result, its gradient the
    # node's incoming gradient
    r = torch.randn(*shape, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    before = ours.running_mean.clone()
    with_res = _act_bn(act, ours, a, residual=r)
    with torch.no_grad():
        ours.running_mean.copy_(before)
    assert with_res.grad_fn.__class__.__name__.startswith("_LeakyBatchNormRows") and torch.equal(with_res, out + r)
    gr, ga3 = torch.autograd.grad((with_res.float() * wt).sum(), (r, a))
    assert torch.equal(gr, wt.to(dtype)) and torch.equal(ga3, ga)
    assert _act_bn(act, ours, a, residual=r.detach().float() if dtype == torch.float16 else r.detach().half()).grad_fn.__class__.__name__ == "AddBackward0"
    # evaluation mode / a contiguous map / a hooked module: the modules themselves
    ours.eval()
    assert torch.equal(_act_bn(act, ours, x), ours(act(x))) and torch.equal(_act_bn(act, ours, x, residual=r.detach()), r.detach() + ours(act(x)))


@pytest.mark.parametrize("dilation", [1, 2])
@pytest.mark.parametrize("shape", [(2, 32, 19, 45), (1, 32, 4, 32), (3, 32, 64, 96)])
def test_conv3x3_c32_rows_equals_the_module(shape, dilation):
    """UNet2D's 3 x 3, 32 -> 32 channel layers on channels-last half maps (csrc/conv2d_rows.hip, unet2d._conv) against nn.Conv2d
    evaluated in float64 on the same half inputs: output, input gradient, weight and bias gradient (all ours); sizes that are no multiple of the 32-pixel segments / 4-row tiles; run-to-run identical"""
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import _conv
    g = torch.Generator().manual_seed(13)
    conv = torch.nn.Conv2d(32, 32, 3, padding=dilation, dilation=dilation).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(*shape, generator=g).cuda().half().contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = torch.randn(*shape, generator=g).cuda()
    with torch.autocast("cuda", dtype=torch.float16):
        y = _conv(conv, x)
    assert y.dtype == torch.float16 and y.grad_fn.__class__.__name__.startswith("_Conv3x3C32Rows") and y.is_contiguous(memory_format=torch.channels_last)
    gx, gw, gb = torch.autograd.grad((y.float() * wt).sum(), (x, conv.weight, conv.bias), retain_graph=True)
    gx2, gw2, gb2 = torch.autograd.grad((y.float() * wt).sum(), (x, conv.weight, conv.bias))
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2) and gw.dtype == torch.float32 and gb.dtype == torch.float32
    ref = torch.nn.Conv2d(32, 32, 3, padding=dilation, dilation=dilation).cuda().double()
    with torch.no_grad():
        ref.weight.copy_(conv.weight.half().double())          # the half weight autocast hands the kernel
        ref.bias.copy_(conv.bias.double())
    xd = x.detach().double().requires_grad_()
    want = ref(xd)
    wx, ww, wb = torch.autograd.grad((want * wt.double()).sum(), (xd, ref.weight, ref.bias))
    rel = lambda a, b: float((a.double() - b).norm() / b.norm().clamp_min(1e-30))      # noqa: E731
    assert float((y.double() - want).abs().max()) <= 2e-3 * max(1.0, float(want.abs().max()))      # one rounding to half
    assert rel(gx, wx) <= 2e-3 and rel(gw, ww) <= 5e-3 and rel(gb, wb) <= 5e-3, (rel(gx, wx), rel(gw, ww), rel(gb, wb))
    # a float32 map / another shape of layer: the module itself
    assert _conv(conv, x.detach().float()).grad_fn.__class__.__name__ == "ConvolutionBackward0"
    other = torch.nn.Conv2d(32, 64, 3, padding=1).cuda().half().to(memory_format=torch.channels_last)
    assert torch.equal(_conv(other, x.detach()), other(x.detach()))


@pytest.mark.parametrize("shape,c_out", [((2, 56, 19, 45), 96), ((2, 96, 21, 37), 96), ((1, 32, 9, 33), 64), ((2, 64, 10, 70), 40),
                                         ((1, 8, 8, 32), 8), ((3, 96, 64, 96), 56)])
def test_conv3x3_rows_general_channels_equals_the_module(shape, c_out, monkeypatch):
    """the decoder's wide 3 x 3 layers (UpBlock.conv1: 56 -> 96, 96 -> 96; csrc/conv2d_rows.hip's general kernel behind unet2d._conv)
    against nn.Conv2d evaluated in float64 on the same half inputs: output, input gradient, weight and bias gradient (all ours: the
    9 x C_in x C_out sums in registers, 96 x 96 in two passes); channel counts that are no multiple of 16 / 32, ragged image sizes;
    run-to-run identical"""
    from taseg_amd import backend as B
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet import unet2d
    monkeypatch.setattr(unet2d, "_CONV_ROWS_MIN_PIXELS", 1)
    g = torch.Generator().manual_seed(17)
    c_in = shape[1]
    conv = torch.nn.Conv2d(c_in, c_out, 3, padding=1).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(*shape, generator=g).cuda().half().contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = torch.randn(shape[0], c_out, *shape[2:], generator=g).cuda()
    with torch.autocast("cuda", dtype=torch.float16):
        y = unet2d._conv(conv, x)
    assert y.dtype == torch.float16 and y.grad_fn.__class__.__name__.startswith("_Conv3x3Rows") and y.is_contiguous(memory_format=torch.channels_last)
    gx, gw, gb = torch.autograd.grad((y.float() * wt).sum(), (x, conv.weight, conv.bias), retain_graph=True)
    gx2, gw2, gb2 = torch.autograd.grad((y.float() * wt).sum(), (x, conv.weight, conv.bias))
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2) and gw.dtype == torch.float32 and gb.dtype == torch.float32
    ref = torch.nn.Conv2d(c_in, c_out, 3, padding=1).cuda().double()
    with torch.no_grad():
        ref.weight.copy_(conv.weight.half().double())
        ref.bias.copy_(conv.bias.double())
    xd = x.detach().double().requires_grad_()
    want = ref(xd)
    wx, ww, wb = torch.autograd.grad((want * wt.double()).sum(), (xd, ref.weight, ref.bias))
    rel = lambda a, b: float((a.double() - b).norm() / b.norm().clamp_min(1e-30))      # noqa: E731
    assert float((y.double() - want).abs().max()) <= 2e-3 * max(1.0, float(want.abs().max()))      # one rounding to half
    assert rel(gx, wx) <= 2e-3 and rel(gw, ww) <= 5e-3 and rel(gb, wb) <= 5e-3, (rel(gx, wx), rel(gw, ww), rel(gb, wb))
    # channel counts the kernel does not take: the module itself
    odd = torch.nn.Conv2d(c_in, 20, 3, padding=1).cuda().half().to(memory_format=torch.channels_last)
    assert unet2d._conv(odd, x.detach()).shape[1] == 20 and not B.conv3x3_rows_takes(c_in, 20) and not B.conv3x3_rows_takes(128, 96)
    with pytest.raises(ValueError):
        B.conv3x3_rows(x.detach(), torch.empty(16, dtype=torch.uint8, device="cuda"), None, c_out)


def test_conv3x3_rows_with_more_output_blocks_than_a_workgroup_holds():
    """160 output channels = 5 blocks of 32: two workgroups per tile, the second with one live wave (backend call: unet2d._conv does not
    route such layers - their weight gradient is not built)"""
    from taseg_amd import backend as B
    g = torch.Generator().manual_seed(31)
    conv = torch.nn.Conv2d(32, 160, 3, padding=1).cuda().half().to(memory_format=torch.channels_last)
    x = torch.randn(2, 32, 11, 37, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    got = B.conv3x3_rows(x, B.conv3x3_rows_pack(conv.weight.detach(), 0), conv.bias.detach().float(), 160)
    want = torch.nn.functional.conv2d(x.double(), conv.weight.detach().double(), conv.bias.detach().double(), padding=1)
    assert float((got.double() - want).abs().max()) <= 2e-3 * max(1.0, float(want.abs().max()))
    assert not B.conv3x3_rows_takes(32, 160)


@pytest.mark.parametrize("shape", [(2, 32, 19, 45), (1, 32, 3, 5), (3, 32, 64, 96)])
def test_conv1x1_c32_with_leaky_relu_equals_the_two_modules(shape, monkeypatch):
    """LeakyReLU(Conv2d(32, 32, 1)) as one node (unet2d._conv_act on csrc/conv2d_rows.hip) against the two modules evaluated in float64
    on the same half inputs (the activation applied to the HALF-rounded layer output, as the module pair does under autocast): output,
    input, weight and bias gradient; pixel counts that are no multiple of the 32-pixel segments; run-to-run identical"""
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet import unet2d
    monkeypatch.setattr(unet2d, "_CONV_ROWS_MIN_PIXELS", 1)
    g = torch.Generator().manual_seed(29)
    conv = torch.nn.Conv2d(32, 32, 1).cuda().to(memory_format=torch.channels_last)
    act = torch.nn.LeakyReLU()
    x = torch.randn(*shape, generator=g).cuda().half().contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = torch.randn(*shape, generator=g).cuda()
    with torch.autocast("cuda", dtype=torch.float16):
        y = unet2d._conv_act(conv, act, x)
    assert y.dtype == torch.float16 and y.grad_fn.__class__.__name__.startswith("_Conv1x1C32Act") and y.is_contiguous(memory_format=torch.channels_last)
    gx, gw, gb = torch.autograd.grad((y.float() * wt).sum(), (x, conv.weight, conv.bias), retain_graph=True)
    gx2, gw2, gb2 = torch.autograd.grad((y.float() * wt).sum(), (x, conv.weight, conv.bias))
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2) and gw.dtype == torch.float32 and gb.dtype == torch.float32
    ref = torch.nn.Conv2d(32, 32, 1).cuda().double()
    with torch.no_grad():
        ref.weight.copy_(conv.weight.half().double())
        ref.bias.copy_(conv.bias.double())
    xd = x.detach().double().requires_grad_()
    want = act(ref(xd))
    wx, ww, wb = torch.autograd.grad((want * wt.double()).sum(), (xd, ref.weight, ref.bias))
    rel = lambda a, b: float((a.double() - b).norm() / b.norm().clamp_min(1e-30))      # noqa: E731
    assert float((y.detach().double() - want.detach()).abs().max()) <= 2e-3 * max(1.0, float(want.detach().abs().max()))
    # (an output that rounds to the other side of zero flips a slope: a handful of elements of the input gradient at most)
    assert rel(gx, wx) <= 5e-3 and rel(gw, ww) <= 5e-3 and rel(gb, wb) <= 5e-3, (rel(gx, wx), rel(gw, ww), rel(gb, wb))
    # other layers / float32 maps: the two modules
    assert unet2d._conv_act(conv, act, x.detach().float()).grad_fn.__class__.__name__ == "LeakyReluBackward0"
    wide = torch.nn.Conv2d(32, 64, 1).cuda().half().to(memory_format=torch.channels_last)
    assert torch.equal(unet2d._conv_act(wide, act, x.detach()), act(wide(x.detach())))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape,cs", [((2, 96, 12, 20), 32), ((1, 128, 5, 7), 64), ((3, 256, 3, 4), 128), ((2, 32, 9, 33), 8)])
def test_shuffle_cat_rows_equals_pixel_shuffle_and_cat(shape, cs, dtype):
    """UpBlock's entry as one pass (csrc/shuffle_cat.hip behind unet2d._ShuffleCatRows) against nn.PixelShuffle + torch.cat on the same
    channels-last maps: bit-equal without dropout factors (it is a permutation), one rounding apart with them; the adjoint against
    autograd of the ATen ops; contiguous gradients"""
    from taseg_amd import backend as B
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import _ShuffleCatRows
    g = torch.Generator().manual_seed(23)
    t, c, h, w = shape
    x = torch.randn(*shape, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    skip = torch.randn(t, cs, 2 * h, 2 * w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = torch.randn(t, c // 4 + cs, 2 * h, 2 * w, generator=g).cuda().to(dtype)
    assert B.shuffle_cat_rows_takes(x, skip)
    want = torch.cat((torch.nn.functional.pixel_shuffle(x, 2), skip), dim=1)
    got = _ShuffleCatRows.apply(x, skip, None)
    assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want)
    gx, gs = torch.autograd.grad(got, (x, skip), wt)
    wx, ws = torch.autograd.grad(want, (x, skip), wt)
    assert torch.equal(gx, wx) and torch.equal(gs, ws)
    assert gx.is_contiguous(memory_format=torch.channels_last) and gs.is_contiguous(memory_format=torch.channels_last)
    # with the dropout factors: (x * m1) * m2 in ATen, x * (m1 m2) here
    m = torch.tensor([0.0, 1.25, 1.5625], device="cuda")[torch.randint(0, 3, (t, c // 4 + cs), generator=g).cuda()]
    want = want * m[:, :, None, None].to(dtype)
    got = _ShuffleCatRows.apply(x, skip, m)
    tol = 2e-3 if dtype == torch.float16 else 1e-6
    assert torch.allclose(got.float(), want.float(), rtol=tol, atol=0) and bool(((got == 0) == (want == 0)).all())
    gx, gs = torch.autograd.grad(got, (x, skip), wt)
    wx, ws = torch.autograd.grad(want, (x, skip), wt)
    assert torch.allclose(gx.float(), wx.float(), rtol=tol, atol=0) and torch.allclose(gs.float(), ws.float(), rtol=tol, atol=0)
    # pairs the kernel does not take
    assert not B.shuffle_cat_rows_takes(x, skip.float() if dtype == torch.float16 else skip.half())
    assert not B.shuffle_cat_rows_takes(x.detach().contiguous(), skip)
    with pytest.raises(ValueError):
        B.shuffle_cat_rows_forward(x.detach(), skip.detach()[:, :, :-1])


def test_up_block_fused_entry_equals_the_modules_and_draws_dropout_masks():
    """UpBlock with the fused entry against the module sequence of the reference (unet2d.py:98-115): equal in eval mode; in training the
    folded mask takes the values {0, 1/(1-p), 1/(1-p)^2} (skip channels {0, 1/(1-p)}) at the modules' rates"""
    from taseg_amd.options import options
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d import UpBlock
    torch.manual_seed(5)
    blk = UpBlock(128, 96, 0.2, mid_filters=128 // 4 + 64).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(3, 128, 10, 14, device="cuda").contiguous(memory_format=torch.channels_last)
    skip = torch.randn(3, 64, 20, 28, device="cuda").contiguous(memory_format=torch.channels_last)
    blk.eval()
    with torch.no_grad():
        got = blk(x, skip)
        with options.override(image_shuffle_cat=False):
            want = blk(x, skip)
    assert torch.equal(got, want)
    blk.train()
    m = torch.cat([blk._masks(x.expand(3, -1, -1, -1), skip) for _ in range(200)])          # 600 frames x 96 channels
    up, sk = m[:, :32], m[:, 32:]
    assert set(up.unique().tolist()) <= {0.0, 1.25, 1.5625} and set(sk.unique().tolist()) <= {0.0, 1.25}
    assert abs(float((up == 1.5625).float().mean()) - 0.64) < 0.02 and abs(float((sk == 1.25).float().mean()) - 0.8) < 0.02
    y = blk(x, skip)
    assert y.grad_fn is not None and torch.isfinite(y).all()


def _tiaf_batch(g):
    from taseg_amd.torchsparse import SparseTensor
    dev = "cuda"
    coords = torch.from_numpy(g["coords"]).to(dev)
    fov_coords = torch.from_numpy(g["fov_coords"]).to(dev)
    return {
        "lidar_ms": SparseTensor(torch.from_numpy(g["feats"]).to(dev), coords),
        "targets_ms": SparseTensor(torch.from_numpy(g["labels"]).to(dev), coords),
        "lidar_fov_ms": SparseTensor(torch.from_numpy(g["fov_feats"]).to(dev), fov_coords),
        "image_ms": torch.from_numpy(g["images"]).to(dev),
        "semantic_map_ms": torch.from_numpy(g["semantic"]).to(dev),
        "offset_img": torch.from_numpy(g["offset_img"]).to(dev),
        "offset_ms": torch.tensor([0], device=dev),
    }


def _build_mm():
    from taseg_amd.pcseg.model import build_network
    cfg = make_model_cfg("MinkUNetMsMm", in_dim=5, cr=1.0, num_layer=[1] * 8, **TIAF_CFG)
    return cfg, fill_parameters(build_network(cfg, 20), seed=3).cuda()


def test_tiaf_state_dict_matches_reference(g_minkunet_ms_mm):
    _, model = _build_mm()
    sd = model.state_dict()
    ours = {k: ",".join(map(str, v.shape)) for k, v in sd.items()}
    ref = dict(zip(g_minkunet_ms_mm["state_keys"].tolist(), g_minkunet_ms_mm["state_shapes"].tolist()))
    assert ours == ref          # same names and shapes (registration order of the extra branches differs)


@pytest.mark.parametrize("training", [True, False])
def test_tiaf_model_vs_reference_golden(g_minkunet_ms_mm, training):
    g = g_minkunet_ms_mm
    tag = "train" if training else "eval"
    _, model = _build_mm()
    model.train()
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
        if not training and isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    grabbed = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, key=key: grabbed.__setitem__(key, o.detach().cpu().numpy()))
             for key, m in (("logits", model.classifier), ("fusion_logits", model.classifier_fusion),
                            ("fov_logits", model.lidar_backbone.classifier),
                            ("image_logits", model.image_backbone.classifier))]
    bd = _tiaf_batch(g)
    ret, tb, _ = model(bd)
    for h in hooks:
        h.remove()
    assert int(bd["image_gather_err"]) == 0
    for key in ("image_logits", "fov_logits", "logits", "fusion_logits"):
        assert grabbed[key].shape == g[f"{tag}_{key}"].shape, key
        assert np.abs(grabbed[key] - g[f"{tag}_{key}"]).max() <= LOGIT_TOL, key
    parts = np.array([float(tb[k]) for k in ("loss_lidar", "loss_fusion", "loss_image_s", "loss_image_d",
                                              "loss_image_lidar")])
    assert np.abs(parts - g[f"{tag}_loss_parts"]).max() <= 1e-3
    assert abs(float(tb["loss"]) - float(g[f"{tag}_loss"])) <= 2e-3
    model.zero_grad()
    ret["loss"].backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    tol = 2e-2 if training else 1e-4
    for k in g:
        if k.startswith(f"{tag}_grad/"):
            a, b = grads[k.split("/", 1)[1]].cpu().numpy(), g[k]
            if a.shape != b.shape:
                a = a[..., ::4, ::4]
            # the dense image branch runs on MIOpen (different convolution algorithms than the CPU reference)
            ktol = max(tol, 1e-3) if "image_backbone" in k else tol
            assert np.linalg.norm(a - b) <= ktol * np.linalg.norm(b), k
    names = g[f"{tag}_gradnames"].tolist()
    assert sorted(names) == sorted(grads)
    norms = np.array([float(grads[n].norm()) for n in names])
    assert np.allclose(norms, g[f"{tag}_gradnorms"], rtol=5e-2 if training else 2e-3, atol=1e-6)


def test_tiaf_model_under_autocast_takes_the_row_kernels_and_stays_close(g_minkunet_ms_mm):
    """the mode the reference trains TIAF in (dist_train.sh:18 `--amp`): MinkUNetMsMm under torch.autocast on the golden's inputs - the
    camera branch runs channels-last on the library's kernels (row gathers, average pooling, LeakyReLU + BatchNorm2d, the 3 x 3
    32-channel convolutions), logits and the five losses stay within half-precision distance of the reference's fp32 run"""
    import taseg_amd.pcseg.model.segmentor.voxel.minkunet.unet2d as U
    g = g_minkunet_ms_mm
    seen = set()
    real = {n: getattr(U, n) for n in ("_Conv3x3C32Rows", "_LeakyBatchNormRows", "_AvgPool3s2Rows", "_ImageGatherThrough")}

    def run():
        _, model = _build_mm()
        model.train()
        for m in model.modules():
            if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
                m.eval()
        grabbed = {}
        hooks = [m.register_forward_hook(lambda mod, i, o, key=key: grabbed.__setitem__(key, o.detach().float().cpu().numpy()))
                 for key, m in (("logits", model.classifier), ("fusion_logits", model.classifier_fusion),
                                ("fov_logits", model.lidar_backbone.classifier), ("image_logits", model.image_backbone.classifier))]
        bd = _tiaf_batch(g)
        with torch.autocast("cuda", dtype=torch.float16):
            ret, tb, _ = model(bd)
        for h in hooks:
            h.remove()
        stack, alive = [ret["loss"].grad_fn], []
        while stack:                                   # which of the library's 2-D nodes are in the graph
            node = stack.pop()
            if node is None or id(node) in seen:
                continue
            alive.append(node)                         # (a dropped wrapper's id would be handed to the next one)
            seen.add(id(node))
            seen.add(type(node).__name__)
            stack.extend(fn for fn, _ in node.next_functions)
        model.zero_grad()
        ret["loss"].float().backward()
        grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
        return grabbed, np.array([float(tb[k]) for k in ("loss_lidar", "loss_fusion", "loss_image_s", "loss_image_d", "loss_image_lidar")]), grads

    a, parts_a, grads_a = run()
    names = sorted(t for t in seen if isinstance(t, str))
    for name in real:
        assert any(t.startswith(name) for t in names), (f"{name} is not in the autocast graph", names)
    for key in ("image_logits", "fov_logits", "logits", "fusion_logits"):
        # (half-precision activations through train-mode BatchNorm over a few hundred FOV points: 0.1 on logits of magnitude ~3)
        assert np.abs(a[key] - g[f"train_{key}"]).max() <= 0.2, (key, float(np.abs(a[key] - g[f"train_{key}"]).max()))
    assert np.abs(parts_a - g["train_loss_parts"]).max() <= 5e-2, (parts_a, g["train_loss_parts"])
    # run to run: everything the library computes is bit-reproducible (every kernel's own test asserts it, and the fused-cloud
    # backbone's logits - which the image branch does not reach - are checked here); the image branch as a whole is NOT: the vendor
    # library's fp16 convolution at the deepest level (Conv2d(256, 256, 3) on a 2 x 4 map here) differs by an ulp between two
    # calls on the same input (profiles/r06_image_branch_determinism.txt), and that reaches the FOV and fusion heads
    b, parts_b, _ = run()
    assert np.array_equal(a["logits"], b["logits"])
    for key in ("image_logits", "fov_logits", "fusion_logits"):
        assert np.abs(a[key] - b[key]).max() <= 5e-2, key
    assert np.abs(parts_a - parts_b).max() <= 1e-3
    assert len(grads_a) > 200 and all(torch.isfinite(v).all() for v in grads_a.values())


def test_tiaf_fix_part_param():
    _, model = _build_mm()
    model.fix_part_param()
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    free = [n for n, p in model.named_parameters() if p.requires_grad]
    assert frozen and all(not n.startswith(("image_backbone", "lidar_backbone", "classifier_fusion")) for n in frozen)
    assert free and all(n.startswith(("image_backbone", "lidar_backbone", "classifier_fusion")) for n in free)


def test_tiaf_nuscenes_twin_uses_point_labels(g_minkunet_ms_mm):
    """MinkUNetMsMmNus == MinkUNetMsMm except that the FOV losses read `targets_fov_ms` (minkunet_ms_mm_nus.py:454-455)."""
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    g = g_minkunet_ms_mm
    cfg = make_model_cfg("MinkUNetMsMmNus", in_dim=5, cr=1.0, num_layer=[1] * 8, **TIAF_CFG)
    nus = fill_parameters(build_network(cfg, 20), seed=3).cuda().train()
    _, kitti = _build_mm()
    kitti.train()
    for model in (nus, kitti):
        for m in model.modules():
            if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
                m.eval()
    bd = _tiaf_batch(g)
    _, tb_k, _ = kitti(bd)
    bd2 = _tiaf_batch(g)
    bd2["targets_fov_ms"] = SparseTensor(bd["image_targets_fov"].clone(), bd2["lidar_fov_ms"].C)   # same labels
    _, tb_n, _ = nus(bd2)
    assert abs(float(tb_k["loss"]) - float(tb_n["loss"])) <= 1e-5
    bd3 = _tiaf_batch(g)
    bd3["targets_fov_ms"] = SparseTensor(bd["image_targets_fov"].roll(1), bd3["lidar_fov_ms"].C)   # other labels
    _, tb_z, _ = nus(bd3)
    assert abs(float(tb_z["loss_image_s"]) - float(tb_k["loss_image_s"])) > 1e-3
    assert abs(float(tb_z["loss_image_d"]) - float(tb_k["loss_image_d"])) <= 1e-5      # dense image loss unaffected


# --------------------------------------------------------------------------- mask distillation (MinkUNetMsKd)
def _build_kd():
    from taseg_amd.pcseg.model import build_network
    cfg = make_model_cfg("MinkUNetMsKd", in_dim=5, cr=0.5, num_layer=[1] * 8, SAMPLING_TYPE="random", MAX_VOXEL=100000,
                         FEAT_KD="mse", FEAT_KD_WEIGHT=10.0)
    return cfg, fill_parameters(build_network(cfg, 20), seed=3).cuda()


@pytest.mark.parametrize("training", [True, False])
def test_kd_model_vs_reference_golden(g_minkunet_ms_kd, training):
    """student + frozen teacher + hash-matched feature MSE against the reference's logits (both networks), the two loss
    terms and the student's gradients (tests/golden/model_minkunet_ms_kd.npz)"""
    from taseg_amd.torchsparse import SparseTensor
    g = g_minkunet_ms_kd
    tag = "train" if training else "eval"
    _, model = _build_kd()
    sd = model.state_dict()
    assert {k: ",".join(map(str, v.shape)) for k, v in sd.items()} == \
        dict(zip(g["state_keys"].tolist(), g["state_shapes"].tolist()))
    model.train()
    if not training:
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
    coords, gt_coords = torch.from_numpy(g["coords"]).cuda(), torch.from_numpy(g["gt_coords"]).cuda()
    bd = {"lidar_ms": SparseTensor(torch.from_numpy(g["feats"]).cuda(), coords),
          "targets_ms": SparseTensor(torch.from_numpy(g["labels"]).cuda(), coords),
          "lidar_ms_gt": SparseTensor(torch.from_numpy(g["gt_feats"]).cuda(), gt_coords),
          "offset_ms": torch.tensor([0], device="cuda")}
    grabbed = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, key=key: grabbed.__setitem__(key, o.detach().cpu().numpy()))
             for key, m in (("logits", model.classifier), ("teacher_logits", model.classifier_gt))]
    ret, tb, _ = model(bd)
    for h in hooks:
        h.remove()
    for key in ("teacher_logits", "logits"):
        assert np.abs(grabbed[key] - g[f"{tag}_{key}"]).max() <= LOGIT_TOL, key
    parts = np.array([float(tb["loss_seg"]), float(tb["loss_feat_kd"])])
    assert np.abs(parts - g[f"{tag}_loss_parts"]).max() <= 2e-3 * max(1.0, float(np.abs(g[f"{tag}_loss_parts"]).max()))
    model.zero_grad()
    ret["loss"].backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert not any("_gt" in n for n in grads)                       # the teacher ran without a graph
    assert sorted(grads) == sorted(g[f"{tag}_gradnames"].tolist())
    tol = 2e-2 if training else 1e-4
    for k in g:
        if k.startswith(f"{tag}_grad/"):
            a, b = grads[k.split("/", 1)[1]].cpu().numpy(), g[k]
            assert np.linalg.norm(a - b) <= tol * np.linalg.norm(b), k


def test_kd_checkpoint_loader_fills_the_teacher(tmp_path):
    """a MinkUNetMs checkpoint initialises student AND teacher (minkunet_ms_kd.py:680-717); fix_part_param freezes `_gt`"""
    import logging
    from taseg_amd.pcseg.model import build_network
    src = fill_parameters(build_network(make_model_cfg("MinkUNetMs", in_dim=5, cr=0.5, num_layer=[1] * 8), 20), seed=9)
    path = tmp_path / "ms.pth"
    torch.save({"model_state": {"module." + k: v for k, v in src.state_dict().items()}}, path)
    _, kd = _build_kd()
    kd.load_params_from_file(str(path), logging.getLogger("kd"), to_cpu=True)
    sd, ref = kd.state_dict(), src.state_dict()
    for k, v in ref.items():
        head, _, rest = k.partition(".")
        assert torch.equal(sd[k].cpu(), v) and torch.equal(sd[f"{head}_gt.{rest}"].cpu(), v), k
    kd.fix_part_param()
    assert all(p.requires_grad != ("_gt" in n) for n, p in kd.named_parameters())


def test_kd_feature_loss_on_device_equals_the_per_sample_loop():
    """MinkUNetMsKd._kd_loss_on_device (no host reads) against the reference's literal loop (minkunet_ms_kd.py:617-633): equal where
    nothing is sub-sampled; with MAX_VOXEL below the candidate counts, on features that differ from the teacher's by a constant the
    MSE of ANY subset is that constant squared - and the gradient reaches exactly min(candidates, MAX_VOXEL) rows per sample"""
    _, model = _build_kd()
    torch.manual_seed(5)
    n, nt, c, bs = 5000, 4000, 48, 3
    batch = torch.sort(torch.randint(0, bs, (n,), device="cuda"))[0].int()
    s2t = torch.randint(-1, nt, (n,), device="cuda")
    s2t[torch.rand(n, device="cuda") < 0.3] = -1
    feat_t = torch.randn(nt, c, device="cuda")
    feat_s = torch.randn(n, c, device="cuda", requires_grad=True)

    def loop(max_voxel):
        total = feat_s.new_zeros(())
        for b in range(bs):
            pick = ((s2t >= 0) & (batch == b)).nonzero().reshape(-1)[:max_voxel]
            total = total + torch.nn.functional.mse_loss(feat_s[pick], feat_t[s2t[pick]]) * model.feat_kd_weight / bs
        return total

    model.max_voxel = 100000
    a, b = model._kd_loss_on_device(feat_s, feat_t, s2t, batch, bs), loop(100000)
    assert abs(float(a) - float(b)) <= 1e-5 * abs(float(b))
    ga, = torch.autograd.grad(a, feat_s)
    gb, = torch.autograd.grad(b, feat_s)
    assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-9)

    model.max_voxel = 200
    shifted = (feat_t[s2t.clamp(min=0)] + 0.5).detach().requires_grad_(True)
    loss = model._kd_loss_on_device(shifted, feat_t, s2t, batch, bs)
    assert abs(float(loss) - 0.25 * model.feat_kd_weight) <= 1e-5
    g, = torch.autograd.grad(loss, shifted)
    touched = (g.abs().sum(1) > 0)
    for b in range(bs):
        cand = int(((s2t >= 0) & (batch == b)).sum())
        assert int((touched & (batch == b)).sum()) == min(cand, 200)
    assert not bool((touched & (s2t < 0)).any())
    # a sample without common voxels: NaN like mse_loss of an empty selection
    s2t_empty = torch.where(batch == 1, torch.full_like(s2t, -1), s2t)
    assert torch.isnan(model._kd_loss_on_device(feat_s, feat_t, s2t_empty, batch, bs))
