"""The stage programs (csrc/fastpath/stage_program.h, minkunet/stage_program.py: one C++ issue loop and one autograd node per
encoder / decoder stage) against the per-block nodes they replace: the same backend calls in the same order, so every result is
compared BIT FOR BIT - loss, logits, all 191 gradients, running statistics, parameters after optimizer steps, evaluation logits;
fp32 and autocast; plain autograd delivery and gradient-bucket slots; weight gradients on the caller's stream and on the second
one.  Reference structure: R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:83-129 (blocks), :393-422 (the U-Net pass)."""
import numpy as np
import pytest
import torch

from taseg_amd.data.synthetic import fill_parameters, make_model_cfg, synth_scan

pytestmark = pytest.mark.gpu


def _scan_batch(seed=5, n_points=30000, batch=1):
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    cs, fs, ls = [], [], []
    for b in range(batch):
        pts, lab = synth_scan(seed + 7 * b, n_points=n_points, n_beams=32, n_az=1400)
        pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
        pc -= pc.min(0)
        _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
        cs.append(np.concatenate([pc[idx], np.full((len(idx), 1), b, np.int32)], 1))
        fs.append(pts[idx])
        ls.append(lab[idx].astype(np.int64))
    coords = torch.from_numpy(np.concatenate(cs)).cuda()
    feats = torch.from_numpy(np.concatenate(fs)).cuda()
    labels = torch.from_numpy(np.concatenate(ls)).cuda()

    def make():
        return {"lidar": SparseTensor(feats.clone(), coords), "targets": SparseTensor(labels, coords), "offset": torch.tensor([0])}
    return make, int(coords.shape[0])


def _model(num_layer=None, seed=3, name="MinkUNet"):
    from taseg_amd.pcseg.model import build_network
    cfg = make_model_cfg(name, in_dim=4, cr=1.0, **({"num_layer": num_layer} if num_layer else {}))
    return fill_parameters(build_network(cfg, 20), seed=seed).cuda()


def _set(on):
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet import stage_program as SP
    SP._ON = on
    return SP


def _train_pass(model, make, amp, use_programs):
    SP = _set(use_programs)
    grabbed = {}
    h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach().clone()))
    model.zero_grad(set_to_none=True)
    torch.manual_seed(11)                                  # the two dropout masks
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        ret, _, _ = model(make())
    ret["loss"].float().backward()
    torch.cuda.synchronize()
    h.remove()
    _set(True)
    progs = SP.compiled(model)
    return dict(loss=ret["loss"].detach().clone(), logits=grabbed["logits"],
                grads={k: p.grad.detach().clone() for k, p in model.named_parameters()},
                buffers={k: b.detach().clone() for k, b in model.named_buffers()}, compiled=bool(progs)), SP


@pytest.mark.parametrize("amp", [False, True])
def test_stage_programs_train_step_equals_the_block_nodes(amp):
    make, n = _scan_batch()
    a, b = _model(), _model()
    a.train()
    b.train()
    ref, _ = _train_pass(a, make, amp, use_programs=False)
    got, _ = _train_pass(b, make, amp, use_programs=True)
    assert got["compiled"] and not ref["compiled"]          # the second model really went through the programs
    assert torch.equal(ref["loss"], got["loss"])
    assert torch.equal(ref["logits"], got["logits"])
    assert set(ref["grads"]) == set(got["grads"]) and len(got["grads"]) == 191          # 63 x (kernel, BatchNorm weight, bias) + the class head
    bad = [k for k in ref["grads"] if not torch.equal(ref["grads"][k], got["grads"][k])]
    assert not bad, bad[:5]
    bad = [k for k in ref["buffers"] if not torch.equal(ref["buffers"][k], got["buffers"][k])]
    assert not bad, bad[:5]


@pytest.mark.parametrize("amp", [False, True])
def test_stage_programs_evaluation_equals_the_block_calls(amp):
    make, n = _scan_batch(seed=8)
    model = _model(seed=5).eval()
    out = {}
    for on in (False, True):
        _set(on)
        bd = make()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            plan = model.prepare(bd)
            x = bd["lidar"]
            from taseg_amd.torchsparse.nn import functional as spF
            feats = spF.spvoxelize(x.F, plan["vox_idx"], plan["vox_counts"])
            out[on] = model._unet(feats, x.F, plan).float().clone()
    SP = _set(True)
    assert torch.equal(out[False], out[True])
    assert SP.compiled(model)


@pytest.mark.parametrize("amp,side", [(False, False), (True, False), (False, True)])
def test_stage_programs_with_flat_sgd_keep_the_parameters_bit_equal(amp, side):
    """three optimizer steps on FlatSGD (gradient-bucket slots: the programs write all three gradients of a layer straight into
    them), per-block nodes against stage programs, optionally with the weight gradients on the second stream"""
    from taseg_amd import _fast
    from taseg_amd.optim import FlatSGD
    make, n = _scan_batch(batch=2, n_points=20000)
    res = {}
    for on in (False, True):
        SP = _set(on)
        model = _model(num_layer=[1, 2, 1, 2, 1, 1, 1, 1]).train()
        opt = FlatSGD(model, lr=0.05, momentum=0.9, weight_decay=1e-4, max_norm=10.0, amp=amp)
        _fast.wgrad_stream(side)
        losses = []
        for it in range(3):
            opt.zero_grad()
            torch.manual_seed(100 + it)
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                ret, _, _ = model(make())
            (ret["loss"].float().mean() * opt.loss_scale()).backward()
            opt.step()
            losses.append(float(ret["loss"]))
        torch.cuda.synchronize()
        _fast.wgrad_stream(False)
        res[on] = (losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, SP.compiled(model))
    _set(True)
    assert res[True][2] and not res[False][2]
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    bad = [k for k in res[False][1] if not torch.equal(res[False][1][k], res[True][1][k])]
    assert not bad, bad[:5]


def test_running_statistics_written_by_a_training_pass_reach_the_evaluation_pass():
    """the block calls update running_mean / running_var through raw pointers: an evaluation pass after MORE training must not reuse
    the 1 / sqrt(running_var + eps) it cached before (both paths)"""
    make, n = _scan_batch(seed=9)
    for on in (False, True):
        _set(on)
        model = _model(num_layer=[1] * 8, seed=2)

        def evaluate():
            model.eval()
            bd = make()
            with torch.no_grad():
                plan = model.prepare(bd)
                from taseg_amd.torchsparse.nn import functional as spF
                x = bd["lidar"]
                return model._unet(spF.spvoxelize(x.F, plan["vox_idx"], plan["vox_counts"]), x.F, plan).clone()

        def train_once():
            model.train()
            model.zero_grad(set_to_none=True)
            ret, _, _ = model(make())
            ret["loss"].backward()

        train_once()
        first = evaluate()
        train_once()
        second = evaluate()
        # a model that only ever evaluates with the SAME buffers gives the reference answer for `second`
        fresh = _model(num_layer=[1] * 8, seed=2)
        fresh.load_state_dict(model.state_dict())
        model_backup, model = model, fresh
        want = evaluate()
        model = model_backup
        assert not torch.equal(first, second)
        assert torch.equal(second, want), float((second - want).abs().max())
    _set(True)


def test_stage_program_falls_back_where_it_does_not_apply():
    """hooks on a conv module, a frozen BatchNorm layer: the module-by-module path serves the pass (same results as ever)"""
    make, n = _scan_batch(seed=4)
    model = _model(num_layer=[1] * 8).train()
    seen = []
    h = model.stage2[1].register_forward_hook(lambda m, i, o: seen.append(tuple(o.F.shape)))      # on a ResidualBlock, not on a conv
    ret, _, _ = model(make())
    ret["loss"].backward()
    h.remove()
    assert seen, "the hook on stage2's first convolution must fire"
    model.stage3[1].net[1].eval()                     # one BatchNorm on its running statistics inside a training model
    ret, _, _ = model(make())
    assert torch.isfinite(ret["loss"])


@pytest.mark.parametrize("grouped", [True, False])
@pytest.mark.parametrize("half,probs", [(False, False), (True, False), (False, True)])
def test_fused_evaluation_tail_equals_the_sorted_form(grouped, half, probs):
    """csrc/evaltail.hip (rows per scene + gather + arg-max, two launches) against the tensor form with its three stable sorts
    (minkunet.py:435-455): the same dictionary of arrays; a batch whose index arrays are NOT grouped by scene is detected on the
    device and served by the sorted form; an inverse map outside its scene raises like the reference's indexing"""
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.minkunet import unvoxelise_predictions
    from taseg_amd.torchsparse import SparseTensor
    g = torch.Generator().manual_seed(3)
    n_scenes, classes = 3, 20
    vox_counts, pt_counts = [5000, 7000, 3000], [9000, 12000, 4000]
    b_vox = torch.cat([torch.full((c,), b, dtype=torch.int32) for b, c in enumerate(vox_counts)])
    b_pts = torch.cat([torch.full((c,), b, dtype=torch.int32) for b, c in enumerate(pt_counts)])
    inv = torch.cat([torch.randint(0, vox_counts[b], (c,), generator=g) for b, c in enumerate(pt_counts)])
    labels = torch.randint(0, classes, (sum(pt_counts),), generator=g)
    if not grouped:
        perm_v, perm_p = torch.randperm(len(b_vox), generator=g), torch.randperm(len(b_pts), generator=g)
        # a permuted voxel order changes which row a scene-local index names: remap the inverse map accordingly
        rank = torch.empty_like(perm_v)
        for b in range(n_scenes):
            sel = (b_vox[perm_v] == b).nonzero().view(-1)           # positions (new order) of scene b's voxels, ascending
            old_local = perm_v[sel] - sum(vox_counts[:b])
            rank_b = torch.empty(vox_counts[b], dtype=torch.long)
            rank_b[old_local] = torch.arange(len(sel))
            rank[sum(vox_counts[:b]):sum(vox_counts[:b + 1])] = rank_b
        inv = torch.cat([rank[sum(vox_counts[:b]):sum(vox_counts[:b + 1])][inv[sum(pt_counts[:b]):sum(pt_counts[:b + 1])]]
                         for b in range(n_scenes)])
        b_vox, b_pts, inv, labels = b_vox[perm_v], b_pts[perm_p], inv[perm_p], labels[perm_p]
    out = torch.randn(len(b_vox), classes, generator=g)
    if not grouped:
        out = out          # rows follow the (permuted) voxel order by construction of `rank`
    out = (out.half() if half else out).cuda()
    cv = torch.zeros((len(b_vox), 4), dtype=torch.int32)
    cv[:, 3] = b_vox
    cp = torch.zeros((len(b_pts), 4), dtype=torch.int32)
    cp[:, 3] = b_pts
    invs = SparseTensor(inv.cuda(), cp.cuda())
    labs = SparseTensor(labels.cuda(), cp.cuda())
    num_points = torch.tensor([c - 7 for c in pt_counts])          # the scans' own point counts trim the arrays
    names = [f"s{b}" for b in range(n_scenes)]
    a = unvoxelise_predictions(out, cv.cuda()[:, -1], invs, labs, num_points, probs, names=names, _fused=True)
    b = unvoxelise_predictions(out, cv.cuda()[:, -1], invs, labs, num_points, probs, names=names, _fused=False)
    assert a["name"] == b["name"]
    for key in ("point_predict", "point_labels", "point_predict_logits"):
        assert len(a[key]) == len(b[key])
        for x, y in zip(a[key], b[key]):
            assert x.shape == y.shape and np.array_equal(x, y), key
    if grouped:
        bad = inv.clone()
        bad[5] = vox_counts[0] + 3
        with pytest.raises(IndexError):
            unvoxelise_predictions(out, cv.cuda()[:, -1], SparseTensor(bad.cuda(), cp.cuda()), labs, num_points, probs, names=names)


def test_direct_gradient_delivery_counts_the_buckets_down_like_the_hooks():
    """with a reducer behind the parameters the stage programs deliver a stage's gradients themselves - one
    GradBucketReducer.deliver call per stage instead of one AccumulateGrad node + hook per parameter: p.grad IS the bucket view,
    every bucket is launched exactly once, and a second backward pass before finish() raises as it does through the hooks"""
    from taseg_amd.optim import FlatSGD
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet import stage_program as SP
    _set(True)
    make, n = _scan_batch(seed=6)
    model = _model(num_layer=[1] * 8).train()
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=0.0, max_norm=10.0, bucket_mb=8.0)
    calls = []
    real = opt.reducer.deliver
    opt.reducer.deliver = lambda params: (calls.append(len(params)), real(params))[1]
    opt.zero_grad()
    ret, _, _ = model(make())
    ret["loss"].backward()
    n_pairs = sum(len(st.layers) for st in SP.programs_of(model).stages.values())            # conv + BatchNorm pairs inside the 8 stages
    assert SP.compiled(model) and len(calls) == 8 and sum(calls) == 3 * n_pairs, calls
    for b in opt.reducer.buckets:
        for p, v in zip(b["params"], b["views"]):
            assert p.grad is not None and p.grad.data_ptr() == v.data_ptr()
    opt.step()
    torch.cuda.synchronize()
    # a second backward pass inside one optimizer step is refused (the bucket's all-reduce may already be in flight)
    opt.zero_grad()
    ret, _, _ = model(make())
    ret["loss"].backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="one backward pass per finish"):
        ret["loss"].backward()
    opt.reducer.finish()


def test_a_second_forward_pass_before_the_backward_is_refused_under_direct_delivery(monkeypatch):
    """two forward passes of one model with gradients and ONE backward (two views, a consistency loss) under FlatSGD: with direct
    delivery the first pass has taken the parameters off the autograd graph, so the second is refused where it starts (it would
    launch the buckets' all-reduce on partial gradients); with options.direct_grads off autograd sums both passes"""
    from taseg_amd.optim import FlatSGD
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet import stage_program as SP
    _set(True)
    make, n = _scan_batch(seed=8)
    model = _model(num_layer=[1] * 8).train()
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=0.0, max_norm=10.0, bucket_mb=8.0)
    opt.zero_grad()
    ret, _, _ = model(make())
    with pytest.raises(RuntimeError, match="second forward pass"):
        model(make())
    ret["loss"].backward()                   # the first pass is intact
    opt.step()
    torch.cuda.synchronize()
    # the same loop through autograd's AccumulateGrad: both passes contribute
    monkeypatch.setattr(SP, "_DIRECT_GRADS", False)
    model2 = _model(num_layer=[1] * 8).train()
    opt2 = FlatSGD(model2, lr=0.01, momentum=0.9, weight_decay=0.0, max_norm=10.0, bucket_mb=8.0)
    opt2.zero_grad()
    torch.manual_seed(3)
    one, _, _ = model2(make())
    one["loss"].backward()
    g1 = {k: p.grad.detach().clone() for k, p in model2.named_parameters()}
    opt2.reducer.finish()
    opt2.zero_grad()
    torch.manual_seed(3)
    a, _, _ = model2(make())
    torch.manual_seed(3)
    b, _, _ = model2(make())
    (a["loss"] + b["loss"]).backward()
    torch.cuda.synchronize()
    for k, p in model2.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[k], rtol=2e-3, atol=1e-5), k
    opt2.reducer.finish()


@pytest.mark.parametrize("amp", [False, True])
def test_training_mode_pass_without_a_graph_runs_on_the_programs(amp):
    """train-mode BatchNorm under no_grad - the frozen teacher of MinkUNetMsKd (R/.../minkunet_ms_kd.py:533) - on the stage programs:
    per-point features and running statistics equal to the module-by-module pass (fp32: bit for bit), nothing kept for a backward pass"""
    from taseg_amd.torchsparse.nn import functional as spF
    make, n = _scan_batch(seed=9)
    out, stats = {}, {}
    for on in (False, True):
        _set(on)
        model = _model(seed=6).train()
        bd = make()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            plan = model.prepare(bd)
            x = bd["lidar"]
            feats = spF.spvoxelize(x.F, plan["vox_idx"], plan["vox_counts"])
            out[on] = model._unet(feats, x.F, plan).float().clone()
        assert not out[on].requires_grad
        stats[on] = {k: b.detach().clone() for k, b in model.named_buffers()}
        compiled = bool(_set(True).compiled(model))
        assert compiled == on
    if amp:
        # (without a graph the module path stores some intermediate results in another precision than its own training pass, which
        # the programs run: a few half-precision ulps apart)
        scale = float(out[False].abs().max())
        assert float((out[False] - out[True]).abs().max()) <= 8 * 2.0 ** -11 * scale
        for k in stats[False]:
            assert torch.allclose(stats[False][k].float(), stats[True][k].float(), rtol=5e-3, atol=1e-4), k
    else:
        assert torch.equal(out[False], out[True])
        bad = [k for k in stats[False] if not torch.equal(stats[False][k], stats[True][k])]
        assert not bad, bad[:5]
    moved = [k for k in stats[True] if k.endswith("num_batches_tracked") and int(stats[True][k]) != 1]
    assert not moved, moved[:5]                       # every BatchNorm of the pass updated its running statistics once


@pytest.mark.parametrize("training", [True, False])
def test_one_call_backbone_equals_the_stage_by_stage_calls(training):
    """`StagePrograms.run_unet` (stage1 .. up4 in one native call, csrc/fastpath/stage_program.h::unet_run) against eight
    `StagePrograms.run` calls with the dropouts between them in Python: the three feature matrices the point head reads and - in
    training mode - every parameter gradient, bit for bit"""
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.nn import functional as spF
    from taseg_amd.torchsparse import nn as spnn
    make, n = _scan_batch(seed=12)
    SP = _set(True)
    model = _model(seed=8)
    model.train(training)
    got = {}
    for form in ("one call", "stage by stage"):
        bd = make()
        model.zero_grad(set_to_none=True)
        with torch.set_grad_enabled(training):
            plan = model.prepare(bd)
            x = bd["lidar"]
            feats = spF.spvoxelize(x.F, plan["vox_idx"], plan["vox_counts"])
            x0 = SparseTensor(feats, plan["coords"], 1)
            x0.cmaps, x0.kmaps = plan["cmaps"], plan["kmaps"]
            x0 = spnn.conv_bn_act(model.stem[0], model.stem[1], x0, relu=True)
            x0 = spnn.conv_bn_act(model.stem[3], model.stem[4], x0, relu=True)
            progs = model._stage_programs(x0.F, plan)
            assert progs is not None
            f0 = x0.F
            if form == "one call":
                f4, y2, y4 = progs.run_unet(f0, plan, training, model.dropout.p)
            else:
                f1 = progs.run("stage1", (f0,), plan, training)
                f2 = progs.run("stage2", (f1,), plan, training)
                f3 = progs.run("stage3", (f2,), plan, training)
                f4 = progs.run("stage4", (f3,), plan, training)
                y1 = progs.run("up1", (torch.nn.functional.dropout(f4, model.dropout.p, training, False), f3), plan, training)
                y2 = progs.run("up2", (y1, f2), plan, training)
                y3 = progs.run("up3", (torch.nn.functional.dropout(y2, model.dropout.p, training, False), f1), plan, training)
                y4 = progs.run("up4", (y3, f0), plan, training)
            if training:
                (f4.float().sum() + y2.float().sum() * 0.5 + y4.float().sum() * 0.25).backward()
        torch.cuda.synchronize()
        got[form] = dict(outs=[t.detach().clone() for t in (f4, y2, y4)],
                         grads={k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    a, b = got["one call"], got["stage by stage"]
    assert all(torch.equal(x, y) for x, y in zip(a["outs"], b["outs"]))
    assert set(a["grads"]) == set(b["grads"]) and (len(a["grads"]) >= 180 if training else not a["grads"])
    bad = [k for k in a["grads"] if not torch.equal(a["grads"][k], b["grads"][k])]
    assert not bad, bad[:5]
