"""Model-level parity on the GPU: taseg_amd.pcseg MinkUNet / MinkUNetMs (HIP kernels) against the
logits / loss / gradients the REAL reference produced for the same inputs and parameters
(tests/golden/model_*.npz), and against the CPU oracle on a second seeded input.

north_star tolerance: per-point logits within 1e-3 absolute of the reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from taseg_amd.data.synthetic import fill_parameters, make_model_cfg, synth_scan  # noqa: E402

LOGIT_TOL = 1e-3


def _build(name, in_dim):
    from taseg_amd.pcseg.model import build_network
    cfg = make_model_cfg(name, in_dim=in_dim, cr=0.5, num_layer=[1] * 8)
    return cfg, fill_parameters(build_network(cfg, 20), seed=3).cuda()


def _batch(g, key):
    from taseg_amd.torchsparse import SparseTensor
    coords = torch.from_numpy(g["coords"]).cuda()
    bd = {key: SparseTensor(torch.from_numpy(g["feats"]).cuda(), coords)}
    suffix = "" if key == "lidar" else "_ms"
    bd["targets" + suffix] = SparseTensor(torch.from_numpy(g["labels"]).cuda(), coords)
    bd["offset" + suffix] = torch.tensor([0], device="cuda")
    return bd


@pytest.mark.parametrize("name,in_dim,key,fix", [("MinkUNet", 4, "lidar", "g_minkunet"),
                                                 ("MinkUNetMs", 5, "lidar_ms", "g_minkunet_ms")])
@pytest.mark.parametrize("training", [True, False])
def test_model_vs_reference_golden(request, name, in_dim, key, fix, training):
    g = request.getfixturevalue(fix)
    tag = "train" if training else "eval"
    cfg, model = _build(name, in_dim)
    model.train()
    if not training:
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
    grabbed = {}
    h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o))
    ret, tb, _ = model(_batch(g, key))
    h.remove()
    logits = grabbed["logits"].detach().cpu().numpy()
    assert np.abs(logits - g[f"{tag}_logits"]).max() <= LOGIT_TOL
    assert abs(float(tb["loss"]) - float(g[f"{tag}_loss"])) <= 1e-3
    model.zero_grad()
    ret["loss"].backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    tol = 2e-2 if training else 1e-4     # train-mode BN amplifies fp32 ordering noise (see test_oracle_golden)
    for k in g:
        if k.startswith(f"{tag}_grad/"):
            a, b = grads[k.split("/", 1)[1]].cpu().numpy(), g[k]
            assert np.linalg.norm(a - b) <= tol * np.linalg.norm(b), k
    norms = np.array([float(grads[n].norm()) for n, _ in model.named_parameters()])
    assert np.allclose(norms, g[f"{tag}_gradnorms"], rtol=5e-2 if training else 1e-3, atol=1e-6)


def test_state_dict_matches_reference(g_minkunet):
    cfg, model = _build("MinkUNet", 4)
    sd = model.state_dict()
    assert list(sd.keys()) == g_minkunet["state_keys"].tolist()
    assert [",".join(map(str, v.shape)) for v in sd.values()] == g_minkunet["state_shapes"].tolist()


def test_minkunet_vs_oracle_bigger_scan():
    """a 20k-point scan, bs = 2: HIP logits vs the CPU oracle (eval-mode BN), kmaps bit-exact"""
    from oracle import model as OM
    from oracle import ts_oracle as O
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.utils.collate import sparse_collate
    cfg, model = _build("MinkUNet", 4)
    model.eval()
    samples = []
    for seed in (41, 42):
        pts, lab = synth_scan(seed, n_points=20000, n_beams=32, n_az=1000)
        pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
        pc -= pc.min(0)
        idx, _ = O.sparse_quantize(pc)
        samples.append(SparseTensor(torch.from_numpy(pts[idx]), torch.from_numpy(pc[idx])))
    batch = sparse_collate(samples)
    coords, feats = batch.C.int().numpy(), batch.F.float()
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    om = OM.OracleMinkUNet(params, cfg, training=False)
    with torch.no_grad():
        want = om.forward_minkunet(coords, feats).numpy()
    grabbed = {}
    h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o))
    h2 = model.stem[0].register_forward_hook(lambda m, i, o: grabbed.__setitem__("kmaps", o.kmaps))
    x = SparseTensor(feats.cuda(), torch.from_numpy(coords).cuda())
    model.train()            # take the training branch (returns before the eval un-voxelisation) ...
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()         # ... with running statistics
    with torch.no_grad():
        model({"lidar": x, "targets": SparseTensor(torch.zeros(len(coords), dtype=torch.long).cuda(), x.C),
               "offset": torch.tensor([0])})
    h.remove()
    h2.remove()
    got = grabbed["logits"].cpu().numpy()
    assert np.abs(got - want).max() <= LOGIT_TOL
    # rulebooks of the whole pyramid (5 submanifold + 4 strided maps): bit-exact against the oracle's
    assert len(grabbed["kmaps"]) == len(om.debug["kmaps"]) == 9
    for (stride, ks, st), (nbmaps, nbsizes, sizes) in om.debug["kmaps"].items():
        key = ((stride,) * 3, (ks,) * 3, (st,) * 3, (1, 1, 1))
        km = grabbed["kmaps"][key]
        assert km.sizes == sizes
        assert np.array_equal(km.nbmaps.cpu().numpy(), nbmaps)
        assert np.array_equal(km.nbsizes.cpu().numpy(), nbsizes)


def test_eval_branch_unvoxelises(g_minkunet):
    """forward(eval) returns per-point predictions through inverse_map (minkunet.py:435-455)"""
    from taseg_amd.torchsparse import SparseTensor
    cfg, model = _build("MinkUNet", 4)
    model.eval()
    g = g_minkunet
    coords = torch.from_numpy(g["coords"]).cuda()
    n = len(coords)
    rs = np.random.RandomState(0)
    per_scan = [int((g["coords"][:, 3] == b).sum()) for b in range(2)]
    inv = np.concatenate([rs.randint(0, m, size=m + 50) for m in per_scan])
    invc = np.concatenate([np.full((m + 50, 4), b, dtype=np.int32) for b, m in enumerate(per_scan)])
    bd = {"lidar": SparseTensor(torch.from_numpy(g["feats"]).cuda(), coords),
          "inverse_map": SparseTensor(torch.from_numpy(inv).cuda(), torch.from_numpy(invc).cuda()),
          "targets_mapped": SparseTensor(torch.zeros(len(inv), dtype=torch.uint8).cuda(), torch.from_numpy(invc).cuda()),
          "num_points": torch.tensor([m + 50 for m in per_scan]), "name": ["a", "b"]}
    with torch.no_grad():
        out = model(bd)
    assert [len(p) for p in out["point_predict"]] == [m + 50 for m in per_scan]
    assert out["point_predict_logits"][0].shape == (per_scan[0] + 50, 20)
    assert n == sum(per_scan)


@pytest.mark.parametrize("amp", [False, True])
def test_eval_block_call_equals_the_module_chain(g_minkunet, monkeypatch, amp):
    """evaluation (eval-mode BatchNorm, no graph): the one-call block path (ts_conv_block_eval: convolution + one elementwise pass
    on the running statistics) against conv3d -> nn.BatchNorm1d(eval) -> add -> relu chained like the reference's modules
    (minkunet.py:42-51, 117-129); and a BatchNorm buffer that changes is picked up (no stale 1 / sqrt(var + eps))"""
    from taseg_amd.torchsparse.nn import modules as M
    cfg, model = _build("MinkUNet", 4)
    model.train()                       # (the training branch of forward: no inverse maps needed; the BatchNorms are what matters)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    outs = []
    for fused in (True, False, True):
        monkeypatch.setattr(M, "_FUSED_BLOCK", fused)
        grabbed = {}
        h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach().float()))
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            model(_batch(g_minkunet, "lidar"))
        h.remove()
        outs.append(grabbed["logits"])
        if len(outs) == 2:          # new running statistics: the fused path must follow
            with torch.no_grad():
                model.stem[1].running_var.mul_(1.7)
                model.stage2[1].net[1].running_mean.add_(0.05)
    tol = 3e-2 if amp else 2e-5
    assert float((outs[0] - outs[1]).abs().max()) <= tol * max(1.0, float(outs[1].abs().max()))
    assert float((outs[2] - outs[1]).abs().max()) > 5e-3               # the changed buffers changed the logits ...
    monkeypatch.setattr(M, "_FUSED_BLOCK", False)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach().float()))
        model(_batch(g_minkunet, "lidar"))
        h.remove()
    assert float((outs[2] - grabbed["logits"]).abs().max()) <= tol * max(1.0, float(outs[2].abs().max()))      # ... the same way
    # the Python wrapper of the same backend call (taken when taseg_amd/_fast_block.so is absent) gives the same bits
    from taseg_amd import _fast
    if _fast.module() is not None:
        monkeypatch.setattr(M, "_FUSED_BLOCK", True)
        monkeypatch.setattr(_fast, "_mod", None)
        monkeypatch.setattr(_fast, "_tried", True)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("py", o.detach().float()))
            model(_batch(g_minkunet, "lidar"))
            h.remove()
        assert torch.equal(grabbed["py"], outs[2])


def test_eval_tail_in_pass_2_has_the_bits_of_the_separate_pass(tmp_path):
    """ts_conv_block_eval applies the BatchNorm / residual / ReLU tail in the store of pass 2 (TsGatherEpilogue, csrc/conv_pairs*.hip)
    where the convolution ends in its list form.  Same arithmetic as the separate elementwise pass: fp32 logits of a whole evaluation
    pass are bit-equal with TASEG_EVAL_TAIL_IN_PASS2=0 (the switch is read once per process: two child processes); half storage rounds
    once instead of twice and stays within half precision of it."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
from taseg_amd.data.synthetic import fill_parameters, make_model_cfg, synth_scan
from taseg_amd.pcseg.model import build_network
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.utils.quantize import sparse_quantize
model = fill_parameters(build_network(make_model_cfg("MinkUNet", in_dim=4, cr=1.0), 20), seed=5).cuda().train()
torch.manual_seed(0)
for m in model.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.eval()
        with torch.no_grad():
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
pts, lab = synth_scan(2, n_points=40000, n_beams=48, n_az=1200)
pc = np.round(pts[:, :3] / 0.05).astype(np.int32); pc -= pc.min(0)
_, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
coords = torch.from_numpy(np.concatenate([pc[idx], np.zeros((len(idx), 1), np.int32)], 1)).cuda()
bd = lambda: {"lidar": SparseTensor(torch.from_numpy(pts[idx]).cuda(), coords),
              "targets": SparseTensor(torch.from_numpy(lab[idx].astype(np.int64)).cuda(), coords), "offset": torch.tensor([0])}
out = {}
for amp in (False, True):
    grabbed = {}
    h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("y", o.detach().float().cpu().numpy()))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        model(bd())
    h.remove()
    out["amp" if amp else "fp32"] = grabbed["y"]
np.savez(os.environ["OUT"], **out)
print("TAIL_OK")
'''
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("1", "0"):
        path = str(tmp_path / f"tail{mode}.npz")
        env = dict(os.environ, REPO=repo, OUT=path, TASEG_EVAL_TAIL_IN_PASS2=mode)
        run = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert "TAIL_OK" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
        res[mode] = np.load(path)
    assert np.array_equal(res["1"]["fp32"], res["0"]["fp32"]), float(np.abs(res["1"]["fp32"] - res["0"]["fp32"]).max())
    scale = max(1.0, float(np.abs(res["0"]["amp"]).max()))
    assert float(np.abs(res["1"]["amp"] - res["0"]["amp"]).max()) <= 2e-2 * scale
    assert float(np.abs(res["1"]["amp"] - res["1"]["fp32"]).max()) <= 5e-2 * scale       # (and autocast stays autocast-close to fp32)


def test_dropout_does_not_touch_devoxelised_features():
    """DROPOUT_P > 0 (the default when the key is absent is 0.3): z1 / z2 are devoxelised from the features BEFORE
    dropout (minkunet.py:400-412).  The concat path collects its sources before the dropout call, so with the same
    seed it must give the logits of the reference-order path (in-place dropout used to corrupt them)."""
    from taseg_amd.pcseg.model import build_network
    cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, num_layer=[1] * 8, DROPOUT_P=0.3)
    model = fill_parameters(build_network(cfg, 20), seed=3).cuda().train()
    pts, lab = synth_scan(5, n_points=20000, n_beams=32, n_az=1000)
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
    pc -= pc.min(0)
    _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
    coords = torch.from_numpy(np.concatenate([pc[idx], np.zeros((len(idx), 1), np.int32)], 1)).cuda()
    feats = torch.from_numpy(pts[idx]).cuda()
    outs = []
    for concat in (True, False):
        bd = {"lidar": SparseTensor(feats.clone(), coords)}
        plan = model.prepare(bd)
        from taseg_amd.torchsparse.nn import functional as spF
        f0 = spF.spvoxelize(feats, plan["vox_idx"], plan["vox_counts"])
        torch.manual_seed(123)
        blocks = model._unet_point_features(f0, feats, plan, concat=concat)
        z = blocks if concat else torch.cat(blocks, dim=1)
        outs.append(z.detach().clone())
        # gradients flow through both forms
        z.square().mean().backward()
    a, b = outs
    assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-5
    # and dropout really was active: the decoder blocks differ from a no-dropout pass
    cfg0 = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, num_layer=[1] * 8, DROPOUT_P=0.0)
    model0 = fill_parameters(build_network(cfg0, 20), seed=3).cuda().train()
    z0 = model0._unet_point_features(spF.spvoxelize(feats, plan["vox_idx"], plan["vox_counts"]), feats, plan, concat=True)
    c1 = model.classifier[0].in_features
    # z1 (stride-16 encoder output, first block) is taken before any dropout: identical; z3 sees dropped-out inputs
    k1 = int(cfg.PLANES[4] * cfg.cr)
    assert float((z0[:, :k1] - a[:, :k1]).abs().max()) <= 1e-5
    assert float((z0[:, k1:] - a[:, k1:]).abs().max()) > 1e-3 and z0.shape[1] == c1


def test_flat_sgd_leaves_parameters_without_gradient_alone():
    """torch.optim.SGD (the reference's optimizer) skips parameters whose .grad is None - no weight decay, no momentum;
    FlatSGD updates whole buckets and must put such slices back."""
    from taseg_amd.optim import FlatSGD
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({"used": torch.nn.Linear(64, 64), "idle": torch.nn.Linear(64, 64)}).cuda()
    ref = torch.nn.ModuleDict({"used": torch.nn.Linear(64, 64), "idle": torch.nn.Linear(64, 64)}).cuda()
    ref.load_state_dict(net.state_dict())
    opt = FlatSGD(net, lr=0.1, momentum=0.9, weight_decay=0.01)
    topt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=0.01)
    x = torch.randn(32, 64, device="cuda")
    for step in range(3):
        for model, o in ((net, opt), (ref, topt)):
            o.zero_grad(set_to_none=True)
            y = model["used"](x)
            if step == 2:                     # the idle branch joins in the last step (momentum starts from zero there)
                y = y + model["idle"](x)
            y.square().mean().backward()
            o.step()
    for (n, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), n


@pytest.mark.parametrize("mode", ["created", "borrowed"])
def test_syncbn_collective_path_on_one_rank(tmp_path, mode):
    """The SyncBatchNorm / DDP code path that `bench.py --gpus N` executes, on a one-rank RCCL group: fused BN+act
    with the all-reduce between reduction and apply must equal the single-process path, and a DDP-wrapped training
    step must run (the multi-GPU scaling bench cannot be launched from the build box).  mode: the direct path on a communicator
    the library creates (options.rccl_direct = "create") / on the process group's own communicator ("borrow"); both are opt-in,
    the default goes through torch.distributed (third leg of the loop below)."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ["REPO"])
import torch.distributed as dist
from taseg_amd.options import options
options.syncbn_single_rank = True
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
torch.cuda.set_device(0)
from taseg_amd.torchsparse.nn.batchnorm import batch_norm_act_train
g = torch.Generator().manual_seed(0)
n, c = 20011, 96
x = torch.randn(n, c, generator=g).cuda().requires_grad_()
res = torch.randn(n, c, generator=g).cuda().requires_grad_()
w = (torch.rand(c, generator=g) + 0.5).cuda().requires_grad_()
b = torch.randn(c, generator=g).cuda().requires_grad_()
from taseg_amd import rccl
comm = rccl.direct_comm(dist.group.WORLD)
assert comm is not None and comm.value, "no communicator for the direct path"
if os.environ["MODE"] == "borrowed":
    assert id(dist.group.WORLD) in rccl._borrowed
    assert comm.value == dist.group.WORLD._get_backend(torch.device("cuda"))._comm_ptr()
else:
    assert not rccl._borrowed
for dtype, tol in ((torch.float32, 2e-4), (torch.float16, 2e-2)):      # fp32 and half-storage activations
    outs = []
    # single process | SyncBN with the library's own communicator (csrc/rccl.hip) | SyncBN through the process group
    for group, direct in ((None, True), (dist.group.WORLD, True), (dist.group.WORLD, False)):
        rccl._comms[id(dist.group.WORLD)] = comm if direct else None
        xi, ri = x.detach().to(dtype).requires_grad_(), res.detach().to(dtype).requires_grad_()
        rm, rv = torch.zeros(c).cuda(), torch.ones(c).cuda()
        nbt = torch.zeros((), dtype=torch.int64).cuda()
        y = batch_norm_act_train(xi, w, b, rm, rv, 0.1, 1e-5, relu=True, residual=ri, group=group,
                                 num_batches_tracked=nbt)
        assert y.dtype == dtype and int(nbt) == 1
        gx, gr, gw, gb = torch.autograd.grad((y.float() * y.float()).sum(), (xi, ri, w, b))
        outs.append([t.detach().float().cpu() for t in (y, gx, gr, gw, gb, rm, rv)])
    for other in outs[1:]:
        for a, bb in zip(outs[0], other):
            assert torch.allclose(a, bb, rtol=tol, atol=tol), (dtype, float((a - bb).abs().max()))
    for a, bb in zip(outs[1], outs[2]):          # the two transports run the same kernels: identical results
        assert torch.equal(a, bb)
rccl._comms[id(dist.group.WORLD)] = comm
# one DDP + SyncBatchNorm training step of the segmentor
from taseg_amd.data.synthetic import make_model_cfg, synth_scan
from taseg_amd.pcseg.model import build_network
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.utils.quantize import sparse_quantize
cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, num_layer=[1] * 8, if_dist=True)
model = build_network(cfg, 20).cuda().train()
net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], gradient_as_bucket_view=True)
pts, lab = synth_scan(1, n_points=20000, n_beams=32, n_az=1000)
pc = np.round(pts[:, :3] / 0.05).astype(np.int32); pc -= pc.min(0)
_, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
coords = torch.from_numpy(np.concatenate([pc[idx], np.zeros((len(idx), 1), np.int32)], 1)).cuda()
bd = {"lidar": SparseTensor(torch.from_numpy(pts[idx]).cuda(), coords),
      "targets": SparseTensor(torch.from_numpy(lab[idx].astype(np.int64)).cuda(), coords), "offset": torch.tensor([0])}
ret, _, _ = net(bd)
ret["loss"].mean().backward()
assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
rccl.shutdown()
t = torch.ones(4, device="cuda")
dist.all_reduce(t)                      # (a borrowed communicator is still the process group's: not destroyed by shutdown())
assert float(t.sum()) == 4.0
dist.destroy_process_group()
print("SYNC_OK")
'''
    env = dict(os.environ, REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), MODE=mode)
    env["TASEG_RCCL_DIRECT"] = "create" if mode == "created" else "borrow"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "SYNC_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_prefetched_batches_train_like_inline_ones():
    """DevicePrefetcher (index plan of batch i+1 built on the staging stream during step i) must not change results:
    three SGD steps over alternating batches give the same losses and final weights as staging inside forward()."""
    from taseg_amd.data.stage import DevicePrefetcher
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    data = []
    for seed in (3, 4):
        pts, lab = synth_scan(seed, n_points=20000, n_beams=32, n_az=1000)
        pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
        pc -= pc.min(0)
        _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
        coords = torch.from_numpy(np.concatenate([pc[idx], np.zeros((len(idx), 1), np.int32)], 1)).cuda()
        data.append((torch.from_numpy(pts[idx]).cuda(), coords, torch.from_numpy(lab[idx].astype(np.int64)).cuda()))

    def run(prefetch, threaded=False):
        cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, num_layer=[1] * 8)
        model = fill_parameters(build_network(cfg, 20), seed=3).cuda().train()
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        it = [0]

        def make_batch():
            f, c, l = data[it[0] % 2]
            it[0] += 1
            return {"lidar": SparseTensor(f, c), "targets": SparseTensor(l, c), "offset": torch.tensor([0])}

        pf = DevicePrefetcher(make_batch, model.prepare, threaded=threaded) if prefetch else None
        losses = []
        for _ in range(3):
            opt.zero_grad()
            ret, _, _ = model(pf.next() if pf else make_batch())
            ret["loss"].backward()
            opt.step()
            if pf:
                pf.prefetch()
            losses.append(float(ret["loss"]))
        if pf:
            pf.close()
        return losses, [p.detach().clone() for p in model.parameters()]

    la, wa = run(False)
    la2, wa2 = run(False)
    lb, wb = run(True)
    # every kernel of the step is deterministic (the weight gradient sums its partial tiles in a fixed order since round
    # 2): identical runs give identical bits, and staging a batch early on another stream must not change a single one
    assert la == la2 and all(torch.equal(a, b) for a, b in zip(wa, wa2)), "two identical runs differ"
    assert la == lb and all(torch.equal(a, b) for a, b in zip(wa, wb)), "prefetched batches trained differently"
    lc, wc = run(True, threaded=True)       # the stage on a worker thread (bench.py --amp)
    assert la == lc and all(torch.equal(a, b) for a, b in zip(wa, wc)), "batches staged on a worker thread trained differently"


def test_two_batches_staged_ahead_come_back_in_order():
    """DevicePrefetcher(depth=2): two stage jobs in flight, each on a stream and thread of its own; the batches are taken from the
    source one after the other (a generator is not thread-safe) and handed out in source order, every one exactly once."""
    import time
    from taseg_amd.data.stage import DevicePrefetcher
    done = object()

    def source():
        for i in range(7):
            yield {"id": i, "x": torch.full((1000,), float(i), device="cuda")}
    it = source()
    seen_streams = set()

    def prepare(b):
        if b is done:
            return
        time.sleep(0.002 * (1 + b["id"] % 3))                     # uneven stage times: later jobs may finish first
        b["y"] = b["x"] * 2                                       # work on the stage's stream
        b["stream"] = torch.cuda.current_stream().cuda_stream
        seen_streams.add(b["stream"])
    pf = DevicePrefetcher(lambda: next(it, done), prepare, threaded=True, depth=2)
    ids = []
    while True:
        b = pf.next()
        if b is done:
            break
        pf.prefetch_early()
        assert float(b["y"].sum()) == 2000.0 * b["id"]            # (the launch stream waited for the stage's event)
        ids.append(b["id"])
    pf.close()
    assert ids == list(range(7))
    assert len(seen_streams) == 2 and torch.cuda.current_stream().cuda_stream not in seen_streams


def test_miou_parity_200_scans(g_miou):
    """mIoU parity gate of SURVEY 8(d): 200 seeded synthetic scans, identical weights - the HIP model's per-voxel arg-max
    against the REAL reference's (golden: reference MinkUNet run scan by scan on the CPU; here 8 scans per batch through
    the HIP kernels), agreement >= 99.9 %, and the confusion matrix / per-class IoU by the reference's definitions."""
    from taseg_amd import metrics
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, num_layer=[1] * 8)
    model = fill_parameters(build_network(cfg, 20), seed=3)
    with torch.no_grad():
        model.classifier[0].weight.copy_(torch.from_numpy(g_miou["head_weight"]))
        model.classifier[0].bias.copy_(torch.from_numpy(g_miou["head_bias"]))
    model = model.cuda().train()             # training branch (no un-voxelisation) ...
    for m in model.modules():
        if isinstance(m, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.Dropout)):
            m.eval()                         # ... on running statistics, no dropout
    grabbed = {}
    model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o))
    seeds = g_miou["seeds"].tolist()
    hist = torch.zeros((20, 20), dtype=torch.int64, device="cuda")
    preds = []
    for b0 in range(0, len(seeds), 8):
        coords, feats, labels = [], [], []
        for b, seed in enumerate(seeds[b0:b0 + 8]):
            pts, lab = synth_scan(seed, n_points=1500, n_beams=16, n_az=360)
            pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
            pc -= pc.min(0)
            _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
            coords.append(np.concatenate([pc[idx], np.full((len(idx), 1), b, np.int32)], 1))
            feats.append(pts[idx])
            labels.append(lab[idx].astype(np.int64))
        c = torch.from_numpy(np.concatenate(coords)).cuda()
        lab_t = torch.from_numpy(np.concatenate(labels)).cuda()
        with torch.no_grad():
            model({"lidar": SparseTensor(torch.from_numpy(np.concatenate(feats)).cuda(), c),
                   "targets": SparseTensor(lab_t, c), "offset": torch.tensor([0])})
        pred = grabbed["logits"].argmax(1)
        hist += metrics.fast_hist(pred, lab_t, 20)
        preds.append(pred.cpu().numpy().astype(np.uint8))
    got = np.concatenate(preds)
    want = g_miou["pred"]
    assert got.shape == want.shape
    agree = float((got == want).mean())
    same_hist = bool(np.array_equal(hist.cpu().numpy(), g_miou["hist"]))
    iou = metrics.per_class_iu(hist).cpu().numpy()
    print(f"mIoU parity: {len(seeds)} scans, {len(want)} voxels, arg-max agreement {100 * agree:.4f} %, confusion matrix "
          f"identical: {same_hist}, max |IoU - reference| {np.abs(iou - g_miou['iou']).max():.2e}")
    assert agree >= 0.999
    assert np.abs(hist.cpu().numpy() - g_miou["hist"]).sum() <= 2 * (1 - agree) * len(want) + 1e-9
    assert np.abs(iou - g_miou["iou"]).max() <= 1e-3


def test_native_index_plan_equals_python_plan():
    """csrc/fastpath index_plan (coordinate pyramid, 9 kernel maps, trilinear maps, devoxelize orders without the
    interpreter lock) against the Python path it replaces: every tensor bit-identical, same keys, same pair totals."""
    from taseg_amd import _fast
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.minkunet import MinkUNetBackbone
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    if _fast.module() is None:
        pytest.skip("taseg_amd/_fast_block.so has not been built")
    coords = []
    for b, seed in enumerate((7, 8)):
        pts, _ = synth_scan(seed, n_points=20000, n_beams=32, n_az=1000)
        pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
        pc -= pc.min(0)
        _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
        coords.append(np.concatenate([pc[idx], np.full((len(idx), 1), b, np.int32)], 1))
    c = torch.from_numpy(np.concatenate(coords)).cuda()
    native = MinkUNetBackbone._index_plan(c, c.float())
    saved, _fast._mod = _fast._mod, None
    try:
        python = MinkUNetBackbone._index_plan(c, c.float())
    finally:
        _fast._mod = saved
    assert set(native["cmaps"]) == set(python["cmaps"]) and set(native["kmaps"]) == set(python["kmaps"])
    for k in python["cmaps"]:
        assert torch.equal(native["cmaps"][k], python["cmaps"][k]), k
    for k, want in python["kmaps"].items():
        got = native["kmaps"][k]
        assert got.sizes == want.sizes and got.total == want.total, k
        assert torch.equal(got.nbmaps, want.nbmaps) and torch.equal(got.nbsizes, want.nbsizes), k
        for name in ("nbr", "nboffs", "pos_out", "pos_in"):
            assert torch.equal(getattr(got, name), getattr(want, name)), (k, name)
    for name in ("tri_idx", "tri_w", "tri_order"):
        assert set(native[name]) == set(python[name])
        for k in python[name]:
            a, b = native[name][k], python[name][k]
            if isinstance(b, tuple) and isinstance(b[0], str):      # cell-reduced plan of stride 16 (backend.devox_cells)
                assert isinstance(a, tuple) and a[0] == b[0] == "cells" and len(a) == len(b) == 5
                assert all(torch.equal(x, y) for x, y in zip(a[1:4], b[1:4])), (name, k)
                assert torch.equal(a[4][:int(b[3][-1])], b[4][:int(b[3][-1])]), (name, k)
            elif isinstance(b, tuple):     # inverse map (offsets, entries) of strides 1 and 4
                assert isinstance(a, tuple) and torch.equal(a[0], b[0])
                assert torch.equal(a[1][:int(b[0][-1])], b[1][:int(b[0][-1])]), (name, k)
            else:
                assert torch.equal(a, b), (name, k)


def test_conv_block_paths_agree(g_minkunet):
    """conv -> BN -> (+ residual) -> ReLU served three ways - the C++ autograd node (taseg_amd/_fast_block.so), the
    Python node (`functional._ConvBlock`) and the unfused conv3d + bn_act chain - issues the same kernels: one training
    step gives the same loss and (up to the float atomics of the weight gradient) the same gradients."""
    from taseg_amd import _fast
    from taseg_amd.torchsparse.nn import modules as M
    results = []
    for mode in ("native", "python", "unfused"):
        if mode == "native" and _fast.module() is None:
            continue
        saved_mod, saved_flag = _fast._mod, M._FUSED_BLOCK
        try:
            if mode != "native":
                _fast._mod, _fast._tried = None, True
            M._FUSED_BLOCK = mode != "unfused"
            cfg, model = _build("MinkUNet", 4)
            model.train()
            ret, tb, _ = model(_batch(g_minkunet, "lidar"))
            ret["loss"].backward()
            results.append((mode, float(tb["loss"]), [p.grad.detach().clone() for p in model.parameters()],
                            [b.detach().clone() for b in model.buffers()]))
        finally:
            _fast._mod, M._FUSED_BLOCK = saved_mod, saved_flag
    assert len(results) >= 2
    ref = results[-1]                                   # the unfused chain
    for mode, loss, grads, bufs in results[:-1]:
        assert abs(loss - ref[1]) <= 1e-5, (mode, loss, ref[1])
        for a, b in zip(grads, ref[2]):
            assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-7, mode
        for a, b in zip(bufs, ref[3]):                  # BatchNorm running statistics / counters
            assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6), mode


@pytest.mark.parametrize("optimizer", ["torch", "flat"])
def test_presplit_weight_planes_change_nothing(monkeypatch, optimizer):
    """taseg_amd/planes.py: three SGD steps with the pre-split weight planes (the direct-rows pair GEMM reads them in the
    128-wide layers, forward and input gradient; re-split after every optimizer step, FlatSGD announcing its raw-pointer
    update) end in bit-identical losses and weights to three steps without them."""
    from taseg_amd import planes
    from taseg_amd.optim import FlatSGD
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    pts, lab = synth_scan(5, n_points=20000, n_beams=32, n_az=1000)
    pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
    pc -= pc.min(0)
    _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
    coords = torch.from_numpy(np.concatenate([pc[idx], np.zeros((len(idx), 1), np.int32)], 1)).cuda()
    feats, labels = torch.from_numpy(pts[idx]).cuda(), torch.from_numpy(lab[idx].astype(np.int64)).cuda()

    def run(enabled):
        monkeypatch.setattr(planes, "_ENABLED", enabled)
        before = planes.stats["refreshes"]
        cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, num_layer=[1] * 8)
        model = fill_parameters(build_network(cfg, 20), seed=3).cuda().train()
        opt = FlatSGD(model, lr=0.05, momentum=0.9) if optimizer == "flat" else torch.optim.SGD(model.parameters(), lr=0.05)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            ret, _, _ = model({"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords),
                               "offset": torch.tensor([0])})
            ret["loss"].backward()
            opt.step()
            losses.append(float(ret["loss"]))
        model.eval()                      # the unfused path (Conv3d.forward -> conv3d) takes the planes as well
        n = len(coords)
        inv = SparseTensor(torch.arange(n, device="cuda"), coords)
        with torch.no_grad():
            out = model({"lidar": SparseTensor(feats, coords), "inverse_map": inv,
                         "targets_mapped": SparseTensor(torch.zeros(n, dtype=torch.uint8, device="cuda"), coords),
                         "num_points": torch.tensor([n]), "name": ["a"]})
        return (losses, [p.detach().clone() for p in model.parameters()], np.asarray(out["point_predict_logits"][0]),
                planes.stats["refreshes"] - before)

    l0, w0, e0, n0 = run(False)
    l1, w1, e1, n1 = run(True)
    assert n0 == 0 and n1 >= 3 * 4, f"planes were split {n1} times: the mechanism did not engage"
    assert l0 == l1 and all(torch.equal(a, b) for a, b in zip(w0, w1)), "pre-split planes changed the training steps"
    assert np.array_equal(e0, e1), "pre-split planes changed the eval logits"


@pytest.mark.parametrize("node", ["native", "python"])
@pytest.mark.parametrize("inc,outc", [(64, 64), (96, 64)])
def test_residual_block_passes_its_input_through(monkeypatch, node, inc, outc):
    """ResidualBlock hands its input through the first conv block's autograd node (conv_bn_act(passthrough=True)): the
    shortcut's gradient - identity or 1x1x1 conv + BatchNorm - joins the store of conv1's input gradient
    (TsConvBlockOpts.addend) instead of meeting it in an add launch.  Same bits as the separate add, on both nodes."""
    from taseg_amd import _fast
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.minkunet import ResidualBlock
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.nn import modules as M
    if node == "native" and _fast.module() is None:
        pytest.skip("taseg_amd/_fast_block.so has not been built")
    if node == "python":
        monkeypatch.setattr(_fast, "_mod", None)
        monkeypatch.setattr(_fast, "_tried", True)
    rs = np.random.RandomState(inc + outc)
    c = np.unique(rs.randint(0, 24, size=(6000, 3)), axis=0).astype(np.int32)
    coords = torch.from_numpy(np.concatenate([c, np.zeros((len(c), 1), np.int32)], 1)).cuda()
    feats = torch.from_numpy(rs.randn(len(c), inc).astype(np.float32)).cuda()
    gout = torch.from_numpy(rs.randn(len(c), outc).astype(np.float32)).cuda()

    def run(passthrough):
        torch.manual_seed(0)
        block = ResidualBlock(inc, outc).cuda().train()
        if not passthrough:
            real = M.conv_bn_act

            def separate(conv, mod, inp, relu=True, residual=None, passthrough=False):
                out = real(conv, mod, inp, relu=relu, residual=residual)
                return (out, inp) if passthrough else out
            monkeypatch.setattr(M, "conv_bn_act", separate)
            import taseg_amd.torchsparse.nn as spnn
            monkeypatch.setattr(spnn, "conv_bn_act", separate)
        x = feats.clone().requires_grad_()
        y = block(SparseTensor(x, coords, 1))
        y.F.backward(gout)
        monkeypatch.undo()
        if node == "python":
            monkeypatch.setattr(_fast, "_mod", None)
            monkeypatch.setattr(_fast, "_tried", True)
        return y.F.detach().clone(), x.grad.clone(), [p.grad.clone() for p in block.parameters()]

    ya, ga, pa = run(True)
    yb, gb, pb = run(False)
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))


@pytest.mark.parametrize("amp", [False, True])
@pytest.mark.parametrize("node", ["native", "python"])
def test_pointwise_shortcut_block_equals_the_module_chain(monkeypatch, node, amp):
    """The 1x1x1 shortcut of a residual block (minkunet.py:105-111; conv.py:135-140: `feats.matmul(weight)`) + its BatchNorm as ONE
    block call per direction on the identity rulebook (TsConvBlockOpts.natural: the pair GEMM's rows are the result rows) against
    the chained modules (Conv3d -> BatchNorm, torch.matmul reference): training output, all gradients, running statistics, and the
    evaluation form; the C++ node and the Python node issue the same call."""
    from taseg_amd import _fast
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.minkunet import ResidualBlock
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.nn import modules as M
    import taseg_amd.backend as B
    if node == "native" and _fast.module() is None:
        pytest.skip("taseg_amd/_fast_block.so has not been built")

    def pin_node():
        if node == "python":
            monkeypatch.setattr(_fast, "_mod", None)
            monkeypatch.setattr(_fast, "_tried", True)
    inc, outc = 96, 64
    rs = np.random.RandomState(11)
    c = np.unique(rs.randint(0, 28, size=(9000, 3)), axis=0).astype(np.int32)
    coords = torch.from_numpy(np.concatenate([c, np.zeros((len(c), 1), np.int32)], 1)).cuda()
    feats = torch.from_numpy(rs.randn(len(c), inc).astype(np.float32)).cuda()
    gout = torch.from_numpy(rs.randn(len(c), outc).astype(np.float32)).cuda()

    def run(fused, train=True):
        pin_node()
        monkeypatch.setattr(M, "_FUSED_BLOCK", fused)
        torch.manual_seed(0)
        block = ResidualBlock(inc, outc).cuda()
        with torch.no_grad():
            block.downsample[1].weight.uniform_(0.5, 1.5)
            block.downsample[1].bias.uniform_(-0.5, 0.5)
        block.train(train)
        x = feats.clone().requires_grad_(train)
        B.profile_begin(expected_launches=200)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp), torch.set_grad_enabled(train):
            sc = M.conv_bn_act(block.downsample[0], block.downsample[1], SparseTensor(x, coords, 1), relu=False)
        if train:
            sc.F.float().backward(gout)
        recs = B.profile_end()
        bn = block.downsample[1]
        monkeypatch.undo()
        out = [sc.F.detach().float().clone(), bn.running_mean.clone(), bn.running_var.clone()]
        if train:
            out += [x.grad.clone(), block.downsample[0].kernel.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()]
        return out, recs

    for train in (True, False):
        a, ra = run(True, train)
        b, _ = run(False, train)
        # dense reference of the forward: BatchNorm (batch / running statistics) of feats @ W
        torch.manual_seed(0)
        ref_block = ResidualBlock(inc, outc).cuda()
        w = ref_block.downsample[0].kernel.detach().double()
        z = feats.double() @ w
        tol = 2e-2 if amp else 2e-5
        for u, v in zip(a, b):
            assert u.shape == v.shape
            scale = max(1.0, float(v.abs().max()))
            assert float((u - v).abs().max()) <= tol * scale, (train, float((u - v).abs().max()), scale)
        if train:
            mean, var = z.mean(0), z.var(0, unbiased=False)
            assert float((a[1].double() - 0.1 * mean).abs().max()) <= (5e-3 if amp else 1e-5)
            # one pair GEMM per direction, no pass 2, one weight gradient in the fused form
            kinds = sorted(r[0] for r in ra)
            assert kinds == ["conv_wgrad", "pair_gemm", "pair_gemm"], kinds
    # both nodes: same call, same bits
    if node == "native":
        a, _ = run(True, True)
        monkeypatch.setattr(_fast, "_mod", None)
        monkeypatch.setattr(_fast, "_tried", True)
        monkeypatch.setattr(M, "_FUSED_BLOCK", True)
        torch.manual_seed(0)
        block = ResidualBlock(inc, outc).cuda().train()
        with torch.no_grad():
            block.downsample[1].weight.uniform_(0.5, 1.5)
            block.downsample[1].bias.uniform_(-0.5, 0.5)
        x = feats.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            sc = M.conv_bn_act(block.downsample[0], block.downsample[1], SparseTensor(x, coords, 1), relu=False)
        sc.F.float().backward(gout)
        assert torch.equal(sc.F.detach().float(), a[0]) and torch.equal(x.grad, a[3])
        assert torch.equal(block.downsample[0].kernel.grad, a[4])


def test_weight_gradient_bucket_slot_is_handed_out_once_per_step():
    """The conv block's backward writes its weight gradient straight into the parameter's gradient-bucket slot (a full
    overwrite that autograd adopts as p.grad).  That is only sound while nothing has been accumulated for the step: a
    weight used TWICE in one graph and a step WITHOUT zero_grad must still give old + new, not 2 x new."""
    from taseg_amd.parallel import GradBucketReducer
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet.minkunet import BasicConvolutionBlock
    from taseg_amd.torchsparse import SparseTensor
    rs = np.random.RandomState(4)
    c = np.unique(rs.randint(0, 20, size=(5000, 3)), axis=0).astype(np.int32)
    coords = torch.from_numpy(np.concatenate([c, np.zeros((len(c), 1), np.int32)], 1)).cuda()
    feats = torch.from_numpy(rs.randn(len(c), 64).astype(np.float32)).cuda()
    gout = torch.from_numpy(rs.randn(len(c), 64).astype(np.float32)).cuda()

    def grads(with_reducer, passes=1):
        torch.manual_seed(0)
        block = BasicConvolutionBlock(64, 64, ks=3).cuda().train()
        red = GradBucketReducer(block) if with_reducer else None
        for _ in range(passes):                                   # no zero_grad between the passes
            y = block(block(SparseTensor(feats, coords, 1)))      # the same weight twice in one graph
            y.F.backward(gout)
            if red is not None:
                red.finish()
        return [p.grad.detach().clone() for p in block.parameters()]

    want1, got1 = grads(False), grads(True)
    for a, b in zip(got1, want1):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))
    want2, got2 = grads(False, passes=2), grads(True, passes=2)
    for a, b in zip(got2, want2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))
    assert float((want2[0] - want1[0]).abs().max()) > 0           # the second pass really added something


def test_block_call_options_are_explicit_and_checked(monkeypatch):
    """Round 4: what a block call may use beyond its rulebook arrives as an argument (TsConvBlockOpts), nothing is taken from or
    left in thread-local state.  (a) a stale ts_conv_planes_hint of the calling thread does not steer a block call; (b) a class
    plan built from ANOTHER kernel map with the same row count (same K, same n - what the round-3 hint accepted) is ignored: the
    call runs pair GEMM + pass 2 and gives the bits of a call without any plan; (c) the block's own plan is taken (other bits,
    1e-6-close)."""
    from taseg_amd import backend as B
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse import nn as spnn
    from taseg_amd.torchsparse.nn import functional as F
    from taseg_amd import _fast
    monkeypatch.setattr(_fast, "_mod", None)            # the Python node: it builds the options struct from the KernelMap's plans
    monkeypatch.setattr(_fast, "_tried", True)
    for name in ("_CLASS_MIN_ROWS", "_CLASS_MIN_ROWS_96", "_CLASS_MIN_ROWS_128", "_CLASS_MIN_ROWS_HALF"):
        monkeypatch.setattr(F, name, 1)
    rs = np.random.RandomState(5)

    def cloud(seed):                 # a two-voxel-thick random surface: LiDAR-like neighbour masks (the plan's work guard keeps it)
        r = np.random.RandomState(seed)
        xs, ys = np.meshgrid(np.arange(150), np.arange(140), indexing="ij")
        h = np.cumsum(r.randint(-1, 2, size=(150, 140)) * (r.rand(150, 140) < 0.2), axis=1) + 60
        c = np.stack([np.concatenate([xs.ravel(), xs.ravel()]), np.concatenate([ys.ravel(), ys.ravel()]),
                      np.concatenate([h.ravel(), h.ravel() + 1])], 1).astype(np.int32)[:40000]
        return torch.from_numpy(np.concatenate([c, np.zeros((len(c), 1), np.int32)], 1)).cuda()

    ca, cb = cloud(1), cloud(2)
    assert ca.shape == cb.shape and not torch.equal(ca, cb)
    feats = torch.from_numpy(rs.randn(ca.shape[0], 64).astype(np.float32)).cuda()
    gout = torch.from_numpy(rs.randn(ca.shape[0], 64).astype(np.float32)).cuda()
    torch.manual_seed(0)
    conv, bn = spnn.Conv3d(64, 64, 3).cuda(), spnn.BatchNorm(64).cuda().train()

    def run(plan_from=None, own_plan=False, stale_hint=False):
        x = SparseTensor(feats.clone().requires_grad_(), ca, 1)
        F.build_pyramid(x, num_levels=0)
        km = x.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
        if own_plan:
            assert km.build_class_plan() is not None
        if plan_from is not None:                                       # the plan of another cloud's map, forced onto this one
            other = SparseTensor(None, plan_from, 1)
            F.build_pyramid(other, num_levels=0)
            okm = other.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
            assert okm.build_class_plan() is not None
            km.cls = okm.cls
            km._plans.clear()
        if stale_hint:
            junk = torch.zeros(3 * conv.kernel.numel(), dtype=torch.int16, device="cuda")
            B.L.load().ts_conv_planes_hint(B.L.ptr(conv.kernel), B.L.ptr(junk), 27, 64, 64)
        for m in (conv, bn):
            m.zero_grad()
        bn.reset_running_stats()
        y = spnn.conv_bn_act(conv, bn, x, relu=True)
        y.F.backward(gout)
        return y.F.detach().clone(), x.F.grad.clone(), conv.kernel.grad.clone()

    monkeypatch.setattr(F, "_CLASS_GEMM", False)
    base = run()
    hinted = run(stale_hint=True)
    assert all(torch.equal(a, b) for a, b in zip(base, hinted))          # (a)
    monkeypatch.setattr(F, "_CLASS_GEMM", True)
    foreign = run(plan_from=cb)
    assert all(torch.equal(a, b) for a, b in zip(base, foreign))         # (b): the foreign plan was not used
    own = run(own_plan=True)
    assert not torch.equal(own[0], base[0])                              # (c): the class path ran ...
    for a, b in zip(own, base):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))       # ... and computes the same block


def test_weight_gradients_on_a_second_stream_give_the_same_bits():
    """TsConvBlockOpts.wgrad_stream (csrc/block.hip, taseg_amd/_fast.py): the weight gradient of every block on a side stream behind
    an event, its operands in a ring of scratch slots, one join when the backward pass ends - three optimizer steps with it leave
    the same parameters, bit for bit, as three steps without (fp32 and autocast; the deterministic partial-tile sum either way)"""
    from taseg_amd import _fast
    from taseg_amd.optim import FlatSGD
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    if _fast.module() is None:
        pytest.skip("native block node not built")
    import bench
    DEV = "cuda"
    coords, feats, labels, _ = bench.make_scans(3, 2, 20000, "minkunet")
    offset = torch.tensor([len(coords)], device=DEV, dtype=torch.int32)

    def run(side, amp):
        torch.manual_seed(0)
        cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5)
        model = build_network(cfg, 20).to(DEV).train()
        opt = FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=1e-4, max_norm=10.0, amp=amp)
        assert _fast.wgrad_stream(side) == side
        try:
            losses = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                    ret, _, _ = model({"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset})
                (ret["loss"].float().mean() * opt.loss_scale()).backward()
                opt.step()
                losses.append(float(ret["loss"]))
            torch.cuda.synchronize()
        finally:
            _fast.wgrad_stream(False)
        return losses, [p.detach().clone() for p in model.parameters()]

    for amp in (False, True):
        l0, p0 = run(False, amp)
        l1, p1 = run(True, amp)
        assert l0 == l1
        assert all(torch.equal(a, b) for a, b in zip(p0, p1))

    # gradient accumulation: the second pass adds into an existing p.grad on the caller's stream - its blocks keep their weight
    # gradient there (the node is told whether p.grad is undefined when the block runs forward); plain torch optimizer semantics
    def accumulate(side):
        torch.manual_seed(0)
        model = build_network(make_model_cfg("MinkUNet", in_dim=4, cr=0.5), 20).to(DEV).train()
        _fast.wgrad_stream(side)
        try:
            for _ in range(2):
                ret, _, _ = model({"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset})
                ret["loss"].float().mean().backward()
            torch.cuda.synchronize()
        finally:
            _fast.wgrad_stream(False)
        return [p.grad.detach().clone() for p in model.parameters() if p.grad is not None]

    g0, g1 = accumulate(False), accumulate(True)
    assert len(g0) == len(g1) and all(torch.equal(a, b) for a, b in zip(g0, g1))


def test_block_backward_launches_its_weight_gradient_on_a_second_stream_itself(monkeypatch):
    """TsConvBlockOpts.wgrad_stream without wgrad_deferred: ts_conv_block_backward itself enqueues the weight gradient on the second
    stream (event, ring slot, chunk-order sum) - the form a C caller without a launch thread uses.  Driven through the Python
    autograd node with the options struct extended on the way in; the weight gradients of a training step are the bits of the
    one-stream step."""
    from taseg_amd import _fast, backend as B
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    from taseg_amd.torchsparse.nn import functional as F
    import bench
    monkeypatch.setattr(_fast, "_mod", None)            # the Python node (the C++ node defers to its launch thread)
    monkeypatch.setattr(_fast, "_tried", True)
    coords, feats, labels, _ = bench.make_scans(5, 2, 12000, "minkunet")
    offset = torch.tensor([len(coords)], device="cuda", dtype=torch.int32)
    lib = B.L.load()
    side = torch.cuda.Stream()
    ring = [torch.empty(96 << 20, dtype=torch.uint8, device="cuda") for _ in range(8)]
    state = {"on": False, "slot": 0, "used": 0}
    plain = F._block_opts

    def opts_with_side(plan_f, plan_d, planes, w16_current, addend, natural=False):
        o = plain(plan_f, plan_d, planes, w16_current, addend, natural)
        if state["on"] and plan_f is None:              # a backward call (the forward passes its forward plan or builds none: see below)
            slot = state["slot"] = (state["slot"] + 1) % 8
            o.wgrad_stream, o.wgrad_ws, o.wgrad_ws_bytes, o.wgrad_slot = side.cuda_stream, ring[slot].data_ptr(), ring[slot].numel(), slot
            state["used"] += 1
        return o

    monkeypatch.setattr(F, "_block_opts", opts_with_side)

    def grads(on):
        torch.manual_seed(0)
        model = build_network(make_model_cfg("MinkUNet", in_dim=4, cr=0.5), 20).cuda().train()
        ret, _, _ = model({"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset})
        state["on"] = on
        try:
            # (retain_graph: the saved activations the second stream reads stay allocated until it has been joined - what the C++
            # node arranges with record_stream)
            ret["loss"].float().mean().backward(retain_graph=True)
        finally:
            state["on"] = False
        B.L.check(lib.ts_stream_join(B.L.stream(), side.cuda_stream), "ts_stream_join")
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    g0, g1 = grads(False), grads(True)
    assert state["used"] >= 40                               # the blocks' backward calls took the second stream
    assert g0.keys() == g1.keys() and all(torch.equal(g0[k], g1[k]) for k in g0)
