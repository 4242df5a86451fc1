"""TIAF (MinkUNetMsMm) and mask distillation (MinkUNetMsKd) at a size that means something, against the oracle (-m gpu).

The model-level goldens of the two segmentors are toys (3 000 voxels, 633 FOV points, 32 x 64 images, one block per stage).  Here:

* MinkUNetMsMm at mk34 depth on two scans (~100k voxels), >= 20k FOV points projected into four camera frames of 384 x 1280:
  everything below the dense image branch - the image -> point hand-over at three maps (`ts_image_plan`, the through-gathers and
  their in-place adjoints), UNet3D on the FOV cloud, `voxel_to_point_fov` x 3, the MinkUNet on the fused cloud, the fusion head, the
  five losses - against `oracle.model.forward_minkunet_ms_mm` on the reference's own CPU kernels (fp32) with a float64 evaluation as
  the yardstick: four logit sets 1e-3, five loss terms, every sparse parameter's gradient norm at max(1e-3, 2 x the reference
  kernels' own distance to float64), and the gradients that arrive at the three image maps against the adjoint of the gather applied
  to the oracle's row gradients.  The dense 2-D network itself is torch's (MIOpen) and is pinned by the toy golden; its two decoder
  maps are stand-in parameters here, which also keeps MIOpen's solver search for 384 x 1280 stacks out of the suite.
  Reference: R/pcseg/model/segmentor/voxel/minkunet/minkunet_ms_mm.py:442-535, unet2d.py:180-214, unet3d.py:297-316.
* MinkUNetMsKd at mk34 depth on the same clouds fused twice (pseudo-label / ground-truth masks differ on ~10 % of the voxels): both
  networks' logits, the two loss terms and the student's gradient norms against `oracle.model.forward_minkunet_ms_kd`.
  Reference: minkunet_ms_kd.py:532-640.
"""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from taseg_amd.data.synthetic import TIAF_CFG, fill_parameters, make_model_cfg, synth_scan, synth_tiaf_sample  # noqa: E402

LOGIT_TOL = 1e-3
GRAD_TOL = 1e-3
HEIGHT, WIDTH, FRAMES = 384, 1280, 2


def _backend():
    from oracle import model as OM
    return "ref" if os.path.exists(os.path.join(os.path.dirname(OM.__file__), "_ref", "ts_ref_backend.so")) else "numpy"


def _clouds(seeds, n_points):
    """voxelised scans with 5 features (x, y, z, intensity, time flag) as the multi-scan dataset hands them over"""
    from oracle import ts_oracle as O
    out = []
    for s in seeds:
        pts, lab = synth_scan(s, n_points=n_points, n_beams=64, n_az=2000)
        pc = np.round(pts[:, :3] / 0.05).astype(np.int32)
        pc -= pc.min(0)
        idx, _ = O.sparse_quantize(pc)
        feat = np.concatenate([pts, np.ones_like(pts[:, :1])], 1)[idx].astype(np.float32)
        out.append((pc[idx], feat, lab[idx].astype(np.int64)))
    return out


def _norm_bar(name, ours, g32, g64, worst, scale):
    """relative error of a gradient's norm against the float64 evaluation, at max(1e-3, 2 x the reference kernels' own).  A
    parameter whose true gradient is zero (a bias in front of a train-mode BatchNorm) has nothing to be relative to: there every
    evaluation must stay below 1e-6 of the largest gradient norm of the model."""
    n_ours, n_ref, n_64 = float(ours.double().norm()), float(g32.double().norm()), float(g64.double().norm())
    if n_64 <= 1e-9 * scale:
        assert n_ours <= 1e-6 * scale, (name, n_ours, n_64)
        return worst
    e_ours, e_ref = abs(n_ours - n_64) / n_64, abs(n_ref - n_64) / n_64
    assert e_ours <= max(GRAD_TOL, 2.0 * e_ref), (name, e_ours, e_ref)
    return max(worst, (e_ours, e_ref, name))


def test_tiaf_mk34_below_the_image_branch_vs_oracle():
    from oracle import model as OM
    from oracle import ts_oracle as O
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.pcseg.model.segmentor.voxel.minkunet import stage_program as SP
    from taseg_amd.torchsparse import SparseTensor
    clouds = _clouds((21, 22), 75000)
    coords, feats, labels, fov_c, fov_f, sem = [], [], [], [], [], []
    for b, (pc, feat, lab) in enumerate(clouds):
        cam = synth_tiaf_sample(pc, feat, seed=40 + b, frames=FRAMES, height=HEIGHT, width=WIDTH)
        col = np.full((len(pc), 1), b, np.int32)
        coords.append(np.concatenate([pc, col], 1))
        feats.append(feat)
        labels.append(lab)
        fov_c.append(np.concatenate([cam["fov_coords"], col[:len(cam["fov_coords"])]], 1))
        fov_f.append(cam["fov_feats"])
        sem.append(cam["semantic"])
    coords, feats, labels = np.concatenate(coords), np.concatenate(feats), np.concatenate(labels)
    fov_c, fov_f, sem = np.concatenate(fov_c), np.concatenate(fov_f), np.concatenate(sem)
    T = FRAMES * len(clouds)
    offset_img = np.cumsum([FRAMES] * len(clouds)).astype(np.int64)
    assert len(coords) >= 40000 and len(fov_c) >= 20000, (len(coords), len(fov_c))

    cfg = make_model_cfg("MinkUNetMsMm", in_dim=5, cr=1.0, **TIAF_CFG)          # NUM_LAYER [2, 3, 4, 6, 2, 2, 2, 2]
    model = fill_parameters(build_network(cfg, 20), seed=5)
    g = torch.Generator().manual_seed(9)
    img = model.image_backbone
    # the two decoder maps as parameters (the dense network that would produce them is torch's own and stays out); the classifier
    # (a 1 x 1 convolution on the 96-channel map) is the real one, so the full-resolution map has two consumers behind its gather
    img.u2_map = torch.nn.Parameter(0.5 * torch.randn(T, 128, HEIGHT // 4, WIDTH // 4, generator=g))
    img.u4_map = torch.nn.Parameter(0.5 * torch.randn(T, 96, HEIGHT, WIDTH, generator=g))
    img._encode = lambda x: (None, None)
    img._decode_u2 = lambda x5, skips: img.u2_map * 1.0
    img._decode_u4 = lambda u2, skips: img.u4_map * 1.0
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    sparse_names = [n for n, _ in model.named_parameters() if not n.startswith("image_backbone.")]
    dev = "cuda"
    c_dev, fc_dev = torch.from_numpy(coords).to(dev), torch.from_numpy(fov_c).to(dev)
    bd = {"lidar_ms": SparseTensor(torch.from_numpy(feats).to(dev), c_dev),
          "targets_ms": SparseTensor(torch.from_numpy(labels).to(dev), c_dev),
          "lidar_fov_ms": SparseTensor(torch.from_numpy(fov_f).to(dev), fc_dev),
          "image_ms": torch.zeros(T, 3, HEIGHT, WIDTH, device=dev),
          "semantic_map_ms": torch.from_numpy(sem).to(dev),
          "offset_img": torch.from_numpy(offset_img).to(dev), "offset_ms": torch.tensor([0], device=dev)}
    grabbed = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, key=key: grabbed.__setitem__(key, o.detach().float().cpu()))
             for key, m in (("logits", model.classifier), ("fusion_logits", model.classifier_fusion),
                            ("fov_logits", model.lidar_backbone.classifier))]
    ret, tb, _ = model(bd)
    for h in hooks:
        h.remove()
    assert int(bd["image_gather_err"]) == 0
    rows_dev = {k: bd[k].detach().float().cpu() for k in ("image_features_fov", "image_logits_fov")}
    dense_logits = bd["image_logits"].detach().float().cpu()
    model.zero_grad(set_to_none=True)
    ret["loss"].backward()
    torch.cuda.synchronize()
    assert SP.compiled(model), "the stage programs did not serve the fused-cloud backbone"
    ours = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    ours_parts = np.array([float(tb[k]) for k in ("loss_lidar", "loss_fusion", "loss_image_s", "loss_image_d", "loss_image_lidar")])

    # ---- the hand-over itself, bit for bit: gathered rows of the three maps against the oracle's indexing
    pix, pbatch = fov_f[:, -2:], fov_c[:, 3]
    u4_np, u2_np = state["image_backbone.u4_map"].numpy(), state["image_backbone.u2_map"].numpy()
    want_rows = np.concatenate([O.image_gather(u4_np, pix, pbatch, offset_img, 0), O.image_gather(u2_np, pix, pbatch, offset_img, 2)], 1)
    assert np.array_equal(rows_dev["image_features_fov"].numpy(), want_rows)
    assert np.array_equal(rows_dev["image_logits_fov"].numpy(), O.image_gather(dense_logits.numpy(), pix, pbatch, offset_img, 0))
    fov_targets = O.image_gather(sem.astype(np.float32), pix, pbatch, offset_img, 0)[:, 0].astype(np.int64)
    dense_t = np.transpose(sem, (0, 2, 3, 1)).reshape(-1)

    # ---- the oracle twice below the image branch: reference kernels in fp32, numpy in float64
    def oracle_pass(backend, dtype):
        params = {k: v.to(dtype).clone().requires_grad_(k in sparse_names) for k, v in state.items()
                  if v.is_floating_point() and not k.startswith("image_backbone.")}
        rows = rows_dev["image_features_fov"].detach().to(dtype).clone().requires_grad_()
        lrows = rows_dev["image_logits_fov"].detach().to(dtype).clone().requires_grad_()
        dense = dense_logits.detach().to(dtype).permute(0, 2, 3, 1).reshape(-1, 20).clone().requires_grad_()
        out = OM.forward_minkunet_ms_mm(params, cfg, coords, torch.from_numpy(feats).to(dtype), fov_c, torch.from_numpy(fov_f).to(dtype), rows,
                                        lrows, training=True, backend=backend)
        loss, parts = OM.loss_minkunet_ms_mm(out, labels, fov_targets, lrows, dense, dense_t, cfg["LOSS_WEIGHT"], ignore=cfg["IGNORE_LABEL"],
                                             label_smoothing=cfg.get("LABEL_SMOOTHING", 0.0))
        loss.backward()
        return dict(out={k: out[k].detach() for k in ("logits", "fusion_logits", "fov_logits")}, overlap=out["overlap"],
                    parts=np.array([float(p.detach()) for p in parts]), loss=float(loss.detach()),
                    grads={n: params[n].grad.detach() for n in sparse_names if params[n].grad is not None},
                    rows=rows.grad.detach(), lrows=lrows.grad.detach(), dense=dense.grad.detach())

    threads = torch.get_num_threads()
    t0 = time.time()
    torch.set_num_threads(1)
    try:
        o32 = oracle_pass(_backend(), torch.float32)
    finally:
        torch.set_num_threads(threads)
    t1 = time.time()
    o64 = oracle_pass("numpy", torch.float64)
    t2 = time.time()
    n_overlap = int(o64["overlap"].sum())
    assert grabbed["fusion_logits"].shape[0] == n_overlap and n_overlap >= 20000
    worst_logit = 0.0
    for key in ("logits", "fusion_logits", "fov_logits"):
        d = float((grabbed[key].double() - o64["out"][key]).abs().max())
        d32 = float((grabbed[key].double() - o32["out"][key].double()).abs().max())
        assert d <= LOGIT_TOL and d32 <= LOGIT_TOL, (key, d, d32)
        worst_logit = max(worst_logit, d)
    assert np.abs(ours_parts - o64["parts"]).max() <= 1e-3 * max(1.0, float(np.abs(o64["parts"]).max())), (ours_parts, o64["parts"])
    assert abs(float(tb["loss"]) - o64["loss"]) <= 2e-3
    # ---- gradients of every sparse parameter (FOV encoder, fused-cloud MinkUNet, both heads)
    assert sorted(n for n in ours if not n.startswith("image_backbone.")) == sorted(o64["grads"])
    worst = (0.0, 0.0, "")
    scale = max(float(v.norm()) for v in o64["grads"].values())
    for n in o64["grads"]:
        worst = _norm_bar(n, ours[n], o32["grads"][n], o64["grads"][n], worst, scale)
    # ---- what arrives at the image maps: the adjoint of the gather applied to the oracle's row gradients (+ the dense loss and the
    # classifier's share for the full-resolution maps), in float64
    def scatter(rows, channels, shift):
        hs, ws = HEIGHT >> shift, WIDTH >> shift
        flat = torch.zeros(T * hs * ws, channels, dtype=torch.float64)
        start = np.concatenate([[0], offset_img[:-1]])[pbatch]
        addr = ((start + pix[:, 0].astype(np.int64) // HEIGHT) * hs + ((pix[:, 0].astype(np.int64) % HEIGHT) >> shift)) * ws + (pix[:, 1].astype(np.int64) >> shift)
        flat.index_add_(0, torch.from_numpy(addr), rows)
        return flat.view(T, hs, ws, channels).permute(0, 3, 1, 2)

    g_u2 = scatter(o64["rows"][:, 96:], 128, 2)
    got = ours["image_backbone.u2_map"].double()
    assert float((got - g_u2).norm() / g_u2.norm()) <= GRAD_TOL
    g_logits = scatter(o64["lrows"], 20, 0) + o64["dense"].view(T, HEIGHT, WIDTH, 20).permute(0, 3, 1, 2)
    w = state["image_backbone.classifier.0.weight"].double().view(20, 96)
    g_u4 = scatter(o64["rows"][:, :96], 96, 0) + torch.einsum("tchw,cd->tdhw", g_logits, w)
    got = ours["image_backbone.u4_map"].double()
    assert float((got - g_u4).norm() / g_u4.norm()) <= GRAD_TOL
    g_w = torch.einsum("tchw,tdhw->cd", g_logits, state["image_backbone.u4_map"].double())
    got = ours["image_backbone.classifier.0.weight"].double().view(20, 96)
    assert float((got - g_w).norm() / g_w.norm()) <= GRAD_TOL
    print(f"TIAF mk34 below the image branch: {len(coords)} voxels, {len(fov_c)} FOV points ({n_overlap} overlap rows), {T} frames of "
          f"{HEIGHT} x {WIDTH}; oracle passes {t1 - t0:.0f} s (reference kernels, fp32) + {t2 - t1:.0f} s (float64); max |logit - fp64| "
          f"{worst_logit:.2e}; losses {ours_parts.round(5).tolist()} vs {o64['parts'].round(5).tolist()}; gradient norms vs fp64: ours max "
          f"{worst[0]:.2e} (reference kernels there {worst[1]:.2e}, {worst[2]})")


def test_kd_mk34_at_size_vs_oracle():
    from oracle import model as OM
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    rs = np.random.RandomState(3)
    clouds = _clouds((31, 32), 45000)
    coords, feats, labels, gt_coords, gt_feats = [], [], [], [], []
    for b, (pc, feat, lab) in enumerate(clouds):
        # the student's cloud keeps ~90 % of the voxels (pseudo-label masks), the teacher's another ~90 % (ground-truth masks)
        keep_s, keep_t = rs.rand(len(pc)) < 0.9, rs.rand(len(pc)) < 0.9
        col = np.full((len(pc), 1), b, np.int32)
        full = np.concatenate([pc, col], 1)
        coords.append(full[keep_s])
        feats.append(feat[keep_s])
        labels.append(lab[keep_s])
        gt_coords.append(full[keep_t])
        gt_feats.append(feat[keep_t])
    coords, feats, labels = np.concatenate(coords), np.concatenate(feats), np.concatenate(labels)
    gt_coords, gt_feats = np.concatenate(gt_coords), np.concatenate(gt_feats)
    assert len(coords) >= 40000
    cfg = make_model_cfg("MinkUNetMsKd", in_dim=5, cr=1.0, SAMPLING_TYPE="random", MAX_VOXEL=10 ** 7, FEAT_KD="mse", FEAT_KD_WEIGHT=10.0)
    model = fill_parameters(build_network(cfg, 20), seed=6).cuda().train()
    model.fix_part_param()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    learn = [n for n, p in model.named_parameters() if p.requires_grad]
    assert learn and not any("_gt" in n for n in learn)
    c_dev, g_dev = torch.from_numpy(coords).cuda(), torch.from_numpy(gt_coords).cuda()
    bd = {"lidar_ms": SparseTensor(torch.from_numpy(feats).cuda(), c_dev), "targets_ms": SparseTensor(torch.from_numpy(labels).cuda(), c_dev),
          "lidar_ms_gt": SparseTensor(torch.from_numpy(gt_feats).cuda(), g_dev), "offset_ms": torch.tensor([0], device="cuda")}
    grabbed = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, key=key: grabbed.__setitem__(key, o.detach().float().cpu()))
             for key, m in (("logits", model.classifier), ("teacher_logits", model.classifier_gt))]
    ret, tb, _ = model(bd)
    for h in hooks:
        h.remove()
    model.zero_grad(set_to_none=True)
    ret["loss"].backward()
    torch.cuda.synchronize()
    ours = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    assert sorted(ours) == sorted(learn)

    def oracle_pass(backend, dtype):
        params = {k: v.to(dtype).clone().requires_grad_(k in learn) for k, v in state.items() if v.is_floating_point()}
        out = OM.forward_minkunet_ms_kd(params, cfg, coords, torch.from_numpy(feats).to(dtype), gt_coords, torch.from_numpy(gt_feats).to(dtype),
                                        labels, training=True, backend=backend, feat_kd_weight=10.0, ignore=cfg["IGNORE_LABEL"],
                                        label_smoothing=cfg.get("LABEL_SMOOTHING", 0.0))
        out["loss"].backward()
        return dict(logits=out["logits"].detach(), teacher_logits=out["teacher_logits"], parts=np.array([float(out["loss_seg"].detach()),
                    float(out["loss_feat_kd"].detach())]), grads={n: params[n].grad.detach() for n in learn})

    threads = torch.get_num_threads()
    t0 = time.time()
    torch.set_num_threads(1)
    try:
        o32 = oracle_pass(_backend(), torch.float32)
    finally:
        torch.set_num_threads(threads)
    t1 = time.time()
    o64 = oracle_pass("numpy", torch.float64)
    for key in ("teacher_logits", "logits"):
        d = float((grabbed[key].double() - o64[key]).abs().max())
        assert d <= LOGIT_TOL, (key, d)
    parts = np.array([float(tb["loss_seg"]), float(tb["loss_feat_kd"])])
    assert np.abs(parts - o64["parts"]).max() <= 1e-3 * max(1.0, float(np.abs(o64["parts"]).max())), (parts, o64["parts"])
    worst = (0.0, 0.0, "")
    scale = max(float(v.norm()) for v in o64["grads"].values())
    for n in learn:
        worst = _norm_bar(n, ours[n], o32["grads"][n], o64["grads"][n], worst, scale)
    print(f"KD mk34: {len(coords)} student / {len(gt_coords)} teacher voxels; oracle passes {t1 - t0:.0f} s + {time.time() - t1:.0f} s; losses "
          f"{parts.round(5).tolist()} vs {o64['parts'].round(5).tolist()}; gradient norms vs fp64: ours max {worst[0]:.2e} (reference kernels "
          f"there {worst[1]:.2e}, {worst[2]})")
