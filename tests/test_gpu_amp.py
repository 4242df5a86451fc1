"""Half-storage (AMP) path on the GPU - BASELINE configs[4] "fp16 MFMA path", SURVEY.md section 8(d) parity gates for
fp16: report max / mean logit deviation and argmax agreement against the fp32 path (no bit-exact claim)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from taseg_amd.data.synthetic import fill_parameters, make_model_cfg, synth_scan  # noqa: E402


@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize("n,c", [(20011, 96), (4097, 32), (900, 256), (7000, 128), (30011, 64)])
def test_bn_act_half_matches_fp32_kernels(n, c, with_res, relu):
    from taseg_amd.torchsparse.nn.batchnorm import batch_norm_act_train
    g = torch.Generator().manual_seed(n + c)
    x = (torch.randn(n, c, generator=g) * 2 + 0.5).cuda().half()
    res = torch.randn(n, c, generator=g).cuda().half() if with_res else None
    w = (torch.rand(c, generator=g) + 0.5).cuda()
    b = torch.randn(c, generator=g).cuda()
    gy = torch.randn(n, c, generator=g).cuda().half()
    outs = []
    for dtype in (torch.float16, torch.float32):
        xi = x.to(dtype).requires_grad_()
        ri = None if res is None else res.to(dtype).requires_grad_()
        wi, bi = w.clone().requires_grad_(), b.clone().requires_grad_()
        rm, rv, nbt = torch.zeros(c).cuda(), torch.ones(c).cuda(), torch.zeros((), dtype=torch.long).cuda()
        y = batch_norm_act_train(xi, wi, bi, rm, rv, 0.1, 1e-5, relu=relu, residual=ri, num_batches_tracked=nbt)
        assert y.dtype == dtype
        grads = torch.autograd.grad(y, [xi, wi, bi] + ([ri] if ri is not None else []), gy.to(dtype))
        outs.append([y.float(), rm, rv, nbt.float()] + [t.float() for t in grads])
    names = ["y", "running_mean", "running_var", "nbt", "gx", "gw", "gb", "gres"]
    for name, a, bb in zip(names, outs[0], outs[1]):
        scale = max(1.0, float(bb.abs().max()))
        assert float((a - bb).abs().max()) <= 4e-3 * scale, name     # half rounding of y / gx (2^-11 relative)


def _scan_batch(seed=5, n_points=30000):
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    pts, lab = synth_scan(seed, n_points=n_points, n_beams=32, n_az=1400)
    pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
    pc -= pc.min(0)
    _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
    coords = torch.from_numpy(np.concatenate([pc[idx], np.zeros((len(idx), 1), np.int32)], 1)).cuda()

    def make():
        return {"lidar": SparseTensor(torch.from_numpy(pts[idx]).cuda(), coords),
                "targets": SparseTensor(torch.from_numpy(lab[idx].astype(np.int64)).cuda(), coords),
                "offset": torch.tensor([0])}
    return make, len(idx)


def test_minkunet_autocast_vs_fp32():
    """One training step of MinkUNet (mk34 widths) under torch.autocast against the fp32 step: half features and Z,
    fp32 accumulation - logits within a few 1e-2, near-total argmax agreement, gradients aligned."""
    from taseg_amd.pcseg.model import build_network
    cfg = make_model_cfg("MinkUNet", in_dim=4, cr=1.0, num_layer=[1] * 8)
    model = fill_parameters(build_network(cfg, 20), seed=3).cuda().train()
    make, n = _scan_batch()
    res = {}
    for mode in ("fp32", "amp"):
        grabbed = {}
        h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach().float()))
        h2 = model.stage2[1].register_forward_hook(lambda m, i, o: grabbed.__setitem__("feat_dtype", o.F.dtype))
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=(mode == "amp")):
            ret, _, _ = model(make())
        ret["loss"].float().backward()
        h.remove()
        h2.remove()
        res[mode] = (grabbed["logits"], float(ret["loss"]), grabbed["feat_dtype"],
                     {k: p.grad.detach().float().clone() for k, p in model.named_parameters()})
        for m in model.modules():                       # same running statistics for the second pass
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
    (l32, loss32, d32, g32), (l16, loss16, d16, g16) = res["fp32"], res["amp"]
    assert d32 == torch.float32 and d16 == torch.float16          # features really are half inside the network
    dev = (l16 - l32).abs()
    agree = float((l16.argmax(1) == l32.argmax(1)).float().mean())
    print(f"AMP vs fp32 on {n} voxels: max |dlogit| {float(dev.max()):.4f}, mean {float(dev.mean()):.5f}, "
          f"argmax agreement {100 * agree:.2f} %, loss {loss16:.5f} vs {loss32:.5f}")
    assert float(dev.max()) < 0.25 and float(dev.mean()) < 0.02 and agree > 0.99
    assert abs(loss16 - loss32) < 2e-2
    for k in ("stem.0.kernel", "stage2.1.net.0.kernel", "stage4.1.net.3.kernel", "up2.1.0.net.0.kernel", "classifier.0.weight"):
        cos = float(torch.nn.functional.cosine_similarity(g16[k].flatten(), g32[k].flatten(), dim=0))
        assert cos > 0.99, (k, cos)
        assert g16[k].dtype == torch.float32


@pytest.mark.parametrize("amp", [False, True])
def test_flat_sgd_matches_torch_sgd_clip_and_gradscaler(amp):
    """FlatSGD == GradScaler.unscale_ + clip_grad_norm_(max_norm) + SGD(momentum, weight_decay).step + GradScaler.update,
    including a step with non-finite gradients (skipped, loss scale halved) and a growth of the scale."""
    from taseg_amd.optim import FlatSGD

    def net():
        torch.manual_seed(1)
        return torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.ReLU(), torch.nn.Linear(40, 7)).cuda()

    a, b = net(), net()
    ref_opt = torch.optim.SGD(a.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3)
    scaler = torch.amp.GradScaler("cuda", enabled=amp, init_scale=1024.0, growth_interval=3)
    ours = FlatSGD(b, lr=0.05, momentum=0.9, weight_decay=1e-3, max_norm=0.5, amp=amp, init_scale=1024.0,
                   growth_interval=3, bucket_mb=0.002)
    g = torch.Generator().manual_seed(0)
    for it in range(8):
        x = torch.randn(16, 24, generator=g).cuda()
        y = torch.randint(0, 7, (16,), generator=g).cuda()
        # one overflowing step (AMP only: without a GradScaler torch lets NaNs into the weights, FlatSGD skips the step)
        poison = float("inf") if (amp and it == 4) else 1.0
        ref_opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(a(x), y) * poison
        scaler.scale(loss).backward()
        scaler.unscale_(ref_opt)
        torch.nn.utils.clip_grad_norm_(a.parameters(), 0.5)
        scaler.step(ref_opt)
        scaler.update()
        ours.zero_grad()
        loss_b = torch.nn.functional.cross_entropy(b(x), y) * poison
        (loss_b * ours.loss_scale()).backward()
        ours.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6), it
        if amp:
            assert float(ours.state[0]) == float(scaler.get_scale()), it
    if amp:
        assert float(ours.state[0]) != 1024.0               # the schedule moved (backoff at it 4, growth later)
    sd = b.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in zip(sd, b.parameters()))      # parameters still serialise normally
    # every tensor starts on a 64-byte boundary of its flat bucket (the 7-element bias must not misalign its successors:
    # the BatchNorm kernels take parameter pointers and require 16-byte alignment)
    assert all(p.data_ptr() % 64 == 0 and p.grad.data_ptr() % 64 == 0 for p in b.parameters())


@pytest.mark.parametrize("stride,plan_kind", [(1, "csr"), (4, "csr"), (16, "cells")])
def test_half_devoxelize_gives_the_bits_of_the_float_kernels(stride, plan_kind):
    """ts_devoxelize_forward_f16_ld / ts_devoxelize_backward_{csr,cells}_f16_ld: half rows in and out, float32 sums, one
    rounding - bit for bit the float32 kernels between `.float()` and `.half()`, for the single op and for the
    concatenating node, forward and backward."""
    from taseg_amd import backend as B
    from taseg_amd.torchsparse.nn import functional as F
    pts, _ = synth_scan(5, n_points=50000, n_beams=64, n_az=1000)
    pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
    pc -= pc.min(0)
    coords = np.unique(np.concatenate([pc // stride * stride, np.zeros((len(pc), 1), np.int32)], 1), axis=0)
    pcf = torch.from_numpy(np.concatenate([pc.astype(np.float32), np.zeros((len(pc), 1), np.float32)], 1)).cuda()
    idx, w = B.trilinear_map(pcf, torch.from_numpy(coords).cuda(), stride)
    m, n = len(coords), len(pc)
    plan = B.devox_cells(idx, w, m) if plan_kind == "cells" else B.devox_csr(idx, w, m)
    rs = np.random.RandomState(stride)
    for c in (32, 96):
        feat = torch.from_numpy(rs.randn(m, c).astype(np.float32)).cuda().half()
        gout = torch.from_numpy(rs.randn(n, c).astype(np.float32)).cuda().half()
        # the float32 kernels around casts
        want = B.devoxelize_forward_cuda(feat.float(), idx, w).half()
        want_g = B.devoxelize_backward_from(gout.float(), 0, c, idx, w, m, plan).half()
        a = feat.clone().requires_grad_()
        got = F.spdevoxelize(a, idx, w, plan)
        assert got.dtype == torch.float16 and torch.equal(got, want)
        got.backward(gout)
        assert a.grad.dtype == torch.float16 and torch.equal(a.grad, want_g)
    # the concatenating node: two sources into one half matrix, gradient blocks read in place
    f1 = torch.from_numpy(rs.randn(m, 32).astype(np.float32)).cuda().half().requires_grad_()
    f2 = torch.from_numpy(rs.randn(m, 64).astype(np.float32)).cuda().half().requires_grad_()
    out = F.spdevoxelize_cat([f1, f2], [(idx, w, plan), (idx, w, plan)])
    assert out.dtype == torch.float16 and out.shape == (n, 96)
    assert torch.equal(out[:, :32], B.devoxelize_forward_cuda(f1.detach().float(), idx, w).half())
    assert torch.equal(out[:, 32:], B.devoxelize_forward_cuda(f2.detach().float(), idx, w).half())
    g = torch.from_numpy(rs.randn(n, 96).astype(np.float32)).cuda().half()
    out.backward(g)
    assert torch.equal(f1.grad, B.devoxelize_backward_from(g.float(), 0, 32, idx, w, m, plan).half())
    assert torch.equal(f2.grad, B.devoxelize_backward_from(g.float(), 32, 64, idx, w, m, plan).half())
    # without a plan the op keeps the float kernels and still returns half
    b = f1.detach().clone().requires_grad_()
    plain = F.spdevoxelize(b, idx, w)
    assert plain.dtype == torch.float16 and torch.equal(plain, out[:, :32])


def test_kept_half_copy_of_the_weights_follows_the_optimizer():
    """planes.half_for: the half copy the half-storage block calls read in place of a cast per call.  It equals
    weight.half() bit for bit, is refreshed for ALL registered weights by the first request after a version bump
    (torch optimizers) or an invalidate() (FlatSGD's raw-pointer update), 16 weights per launch, and the block call
    that is handed the copy gives the bits of the call that casts for itself."""
    from taseg_amd import planes as P
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse import nn as spnn
    from taseg_amd.torchsparse.nn import modules as M
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(27, ci, co, device="cuda") * 0.1) for ci, co in ((32, 32), (32, 64), (96, 96)) for _ in range(7)]
    before = dict(P.stats)
    hs = [P.half_for(w) for w in ws]
    for w, h in zip(ws, hs):
        assert h.dtype == torch.float16 and torch.equal(h, w.detach().half())
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5)                                   # version bump, same storage
    assert not torch.equal(hs[3], ws[3].detach().half())     # stale until asked for
    batches = P.stats["launch_batches"]
    h0 = P.half_for(ws[0])
    assert h0 is hs[0] and P.stats["launch_batches"] == batches + 1      # one request refreshed every stale weight ...
    for w, h in zip(ws, hs):
        assert torch.equal(h, w.detach().half())
    assert P.half_for(ws[5]) is hs[5] and P.stats["launch_batches"] == batches + 1   # ... and nothing is left to do
    ws[2].data.add_(1.0)                                  # .data writes do not bump the version: invalidate() is the contract
    P.invalidate()
    assert torch.equal(P.half_for(ws[2]), ws[2].detach().half())
    assert P.stats["refreshes"] > before["refreshes"]

    # a conv + BN + ReLU block under autocast: kept copy vs the call's own cast
    rs = np.random.RandomState(0)
    c = np.unique(rs.randint(0, 24, (6000, 3)), axis=0).astype(np.int32)
    coords = torch.from_numpy(np.concatenate([c, np.zeros((len(c), 1), np.int32)], 1)).cuda()
    feats = torch.randn(len(c), 32, device="cuda")
    conv, bn = spnn.Conv3d(32, 64, 3).cuda(), spnn.BatchNorm(64).cuda()

    def run(kept):
        saved = P.half_for
        if not kept:
            P.half_for = lambda w: None
        try:
            bn.reset_running_stats()
            conv.zero_grad(set_to_none=True)
            x = SparseTensor(feats.clone().requires_grad_(True), coords, 1)
            with torch.autocast("cuda", dtype=torch.float16):
                y = M.conv_bn_act(conv, bn, x)
            y.feats.float().square().sum().backward()
            return y.feats.detach().clone(), x.feats.grad.clone(), conv.kernel.grad.clone()
        finally:
            P.half_for = saved

    a, b = run(True), run(False)
    assert a[0].dtype == torch.float16
    for u, v in zip(a, b):
        assert torch.equal(u, v)
