"""Golden-vector generator (build container only; reads /root/reference, never shipped to run).

    python tests/golden/make_golden.py

Runs the REAL reference - its torchsparse v1.4.0 Python + CPU extension and its pcseg model /
dataset code - on small seeded inputs and stores inputs + outputs as .npz fixtures next to
this file.  The fixtures are data only; tests compare the oracle (tests -m "not gpu") and the
HIP path (tests -m gpu) against them.  See _ref_env.py for how the reference is imported and
which two known CPU-backend defects are neutralised.
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import _ref_env  # noqa: E402
from taseg_amd.data.synthetic import (FLEXIBLE_STEPS_KITTI, TIAF_CFG, fill_parameters, make_model_cfg,  # noqa: E402
                                      synth_pose, synth_scan, synth_tiaf_sample)

torch.set_num_threads(1)
BACKEND_DESC = _ref_env.setup_torchsparse()
import torchsparse  # noqa: E402  (the reference library)
import torchsparse.nn.functional as F  # noqa: E402
from torchsparse import SparseTensor  # noqa: E402
from torchsparse.nn.utils import get_kernel_offsets  # noqa: E402
from torchsparse.utils.collate import sparse_collate_fn  # noqa: E402
from torchsparse.utils.quantize import sparse_quantize  # noqa: E402

MinkUNet, MinkUNetMs = _ref_env.setup_pcseg()
VOXEL = 0.05


def small_scan(seed, n=1500, **kw):
    return synth_scan(seed, n_points=n, n_beams=16, n_az=360, **kw)


def dataset_voxelize(points, labels):
    """The reference's own lines (semantickitti_voxel.py:119-137) through its library calls."""
    pc_ = np.round(points[:, :3] / VOXEL).astype(np.int32)
    pc_ -= pc_.min(0, keepdims=1)
    _, inds, inverse = sparse_quantize(pc_, return_index=True, return_inverse=True)
    return pc_, inds, inverse


def make_batch(seeds, in_dim=4):
    samples, raw = [], []
    for s in seeds:
        pts, lab = small_scan(s)
        pc_, inds, inverse = dataset_voxelize(pts, lab)
        feat = pts if in_dim == 4 else np.concatenate([pts, np.ones_like(pts[:, :1])], 1)
        samples.append({"lidar": SparseTensor(feat[inds], pc_[inds]), "targets": SparseTensor(lab[inds], pc_[inds])})
        raw.append((pts, lab, pc_, inds, inverse))
    batch = sparse_collate_fn(samples)
    return batch, raw


# ----------------------------------------------------------------------------------------- op level
def gen_ops():
    out = {"backend": np.array(BACKEND_DESC)}
    batch, raw = make_batch([11, 12])
    coords = batch["lidar"].C.int().contiguous()            # [N,4] x,y,z,b  (bs = 2)
    feats = batch["lidar"].F.float()
    n = coords.shape[0]
    out["coords"] = coords.numpy()
    out["hash"] = F.sphash(coords).numpy()
    off3 = get_kernel_offsets(3, 1, 1)
    out["offsets_k3s1"] = off3.numpy()
    out["khash_k3s1"] = F.sphash(coords, off3).numpy()
    # negative coordinates hash through the (unsigned) cast
    neg = coords.clone()
    neg[:, :3] -= 40
    out["coords_neg"] = neg.numpy()
    out["hash_neg"] = F.sphash(neg).numpy()

    # kernel maps exactly as conv3d builds them (conv.py:156-176), strides 1 -> 2 -> 4 -> 8
    x = SparseTensor(feats, coords, 1)
    cur = coords
    ts = 1
    for level in range(3):
        refs = F.sphash(cur)
        off = get_kernel_offsets(3, ts, 1)
        res = F.sphashquery(F.sphash(cur, off), refs)
        nbsizes = torch.sum(res != -1, dim=1)
        nbmaps = torch.nonzero(res != -1)
        nbmaps[:, 0] = res.view(-1)[nbmaps[:, 0] * res.size(1) + nbmaps[:, 1]]
        out[f"k3_s{ts}_results"] = res.numpy().astype(np.int32)
        out[f"k3_s{ts}_nbmaps"] = nbmaps.numpy().astype(np.int32)
        out[f"k3_s{ts}_nbsizes"] = nbsizes.numpy().astype(np.int32)
        down = F.spdownsample(cur, 2, 2, ts)
        out[f"down_s{ts}"] = down.numpy()
        off2 = get_kernel_offsets(2, ts, 1)
        res2 = F.sphashquery(F.sphash(down, off2), refs)
        nbsizes2 = torch.sum(res2 != -1, dim=1)
        nbmaps2 = torch.nonzero(res2 != -1)
        nbmaps2[:, 0] = res2.view(-1)[nbmaps2[:, 0] * res2.size(1) + nbmaps2[:, 1]]
        out[f"k2_s{ts}_results"] = res2.numpy().astype(np.int32)
        out[f"k2_s{ts}_nbmaps"] = nbmaps2.numpy().astype(np.int32)
        out[f"k2_s{ts}_nbsizes"] = nbsizes2.numpy().astype(np.int32)
        cur, ts = down, ts * 2

    # convolution forward / backward, plain and transposed, through the reference autograd Function
    g = torch.Generator().manual_seed(5)
    for tag, (ci, co) in {"a": (5, 16), "b": (16, 32)}.items():
        xin = (torch.randn(n, ci, generator=g)).requires_grad_()
        w = (torch.randn(27, ci, co, generator=g) * 0.2).requires_grad_()
        st = SparseTensor(xin, coords, 1)
        y = F.conv3d(st, w, 3)
        gy = torch.randn(y.F.shape, generator=g)
        y.F.backward(gy)
        out.update({f"conv_{tag}_x": xin.detach().numpy(), f"conv_{tag}_w": w.detach().numpy(),
                    f"conv_{tag}_y": y.F.detach().numpy(), f"conv_{tag}_gy": gy.numpy(),
                    f"conv_{tag}_gx": xin.grad.numpy(), f"conv_{tag}_gw": w.grad.numpy()})
    # strided + transposed pair (k2 s2 down, then its mirror up) - conv.py:184-192
    xin = torch.randn(n, 16, generator=g).requires_grad_()
    wd = (torch.randn(8, 16, 32, generator=g) * 0.2).requires_grad_()
    wu = (torch.randn(8, 32, 16, generator=g) * 0.2).requires_grad_()
    st = SparseTensor(xin, coords, 1)
    st.cmaps[st.stride] = coords
    yd = F.conv3d(st, wd, 2, stride=2)
    yu = F.conv3d(yd, wu, 2, stride=2, transposed=True)
    gy = torch.randn(yu.F.shape, generator=g)
    yu.F.backward(gy)
    out.update({"convt_x": xin.detach().numpy(), "convt_wd": wd.detach().numpy(), "convt_wu": wu.detach().numpy(),
                "convt_yd": yd.F.detach().numpy(), "convt_yu": yu.F.detach().numpy(), "convt_gy": gy.numpy(),
                "convt_gx": xin.grad.numpy(), "convt_gwd": wd.grad.numpy(), "convt_gwu": wu.grad.numpy(),
                "convt_coords_d": yd.C.numpy()})

    # initial_voxelize pieces + voxelize fwd/bwd on raw (non-unique) points (minkunet/utils.py:11-36)
    pts = np.concatenate([np.concatenate([r[0], np.full((len(r[0]), 1), b, np.float32)], 1)
                          for b, r in enumerate(raw)], 0)
    pc = torch.from_numpy(pts[:, [0, 1, 2, 4]].copy())
    pc[:, :3] = torch.from_numpy(np.round(pts[:, :3] / VOXEL))          # integer-valued float coords, bs = 2
    pc[:, :3] -= pc[:, :3].min(0).values
    pf = torch.from_numpy(pts[:, :4].copy()).requires_grad_()
    pc_hash = F.sphash(torch.floor(pc).int())
    sparse_hash = torch.unique(pc_hash)
    idx_query = F.sphashquery(pc_hash, sparse_hash)
    counts = F.spcount(idx_query.int(), len(sparse_hash))
    vox_c = torch.round(F.spvoxelize(torch.floor(pc), idx_query, counts)).int()
    vox_f = F.spvoxelize(pf, idx_query, counts)
    gv = torch.randn(vox_f.shape, generator=g)
    vox_f.backward(gv)
    out.update({"iv_points_c": pc.numpy(), "iv_points_f": pf.detach().numpy(), "iv_hash": pc_hash.numpy(),
                "iv_sparse_hash": sparse_hash.numpy(), "iv_idx_query": idx_query.numpy().astype(np.int32),
                "iv_counts": counts.numpy(), "iv_vox_c": vox_c.numpy(), "iv_vox_f": vox_f.detach().numpy(),
                "iv_gv": gv.numpy(), "iv_gf": pf.grad.numpy()})

    # voxel_to_point lookup + trilinear weights + devoxelize fwd/bwd, strides 1 and 4, off-grid points
    jitter = torch.rand(pc.shape[0], 3, generator=g) * 0.999
    zc = pc.clone()
    zc[:, :3] = torch.floor(pc[:, :3]) + jitter
    out["tri_points"] = zc.numpy()
    for s, vox in ((1, vox_c), (4, torch.from_numpy(out["down_s2"]))):
        offk = get_kernel_offsets(2, s, 1)
        old_hash = F.sphash(torch.cat([torch.floor(zc[:, :3] / s).int() * s, zc[:, -1].int().view(-1, 1)], 1), offk)
        idxq = F.sphashquery(old_hash, F.sphash(vox.int()))
        wts = F.calc_ti_weights(zc, idxq, scale=s).transpose(0, 1).contiguous()
        idxq = idxq.transpose(0, 1).contiguous()
        vf = torch.randn(vox.shape[0], 12, generator=g).requires_grad_()
        dv = F.spdevoxelize(vf, idxq, wts)
        gd = torch.randn(dv.shape, generator=g)
        dv.backward(gd)
        out.update({f"tri_s{s}_vox": vox.numpy().astype(np.int32), f"tri_s{s}_idx": idxq.numpy().astype(np.int32),
                    f"tri_s{s}_w": wts.numpy(), f"tri_s{s}_feat": vf.detach().numpy(),
                    f"tri_s{s}_out": dv.detach().numpy(), f"tri_s{s}_gout": gd.numpy(),
                    f"tri_s{s}_gfeat": vf.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops.npz:", n, "voxels;", {k: v.shape for k, v in out.items() if "nbmaps" in k})


# ----------------------------------------------------------------------------------------- model level
def run_model(cls, name, in_dim, key, seeds, training):
    cfg = make_model_cfg(name, in_dim=in_dim, cr=0.5, num_layer=[1] * 8)
    torch.manual_seed(0)
    model = cls(cfg, 20)
    fill_parameters(model, seed=3)
    model.train(training)
    batch, raw = make_batch(seeds, in_dim=in_dim)
    lidar = batch["lidar"]
    lidar.F = lidar.F.float()
    lidar.C = lidar.C.int()
    bd = {key: lidar}
    tkey = "targets" if key == "lidar" else "targets_ms"
    bd[tkey] = batch["targets"]
    bd["offset" if key == "lidar" else "offset_ms"] = torch.tensor([0])
    coords_in = lidar.C.numpy().copy()
    feats_in = lidar.F.numpy().copy()
    labels = batch["targets"].F.numpy().astype(np.int64)
    captured = {}
    handle = model.classifier.register_forward_hook(lambda m, i, o: captured.__setitem__("logits", o))
    model.train()  # the training branch returns the loss; BN mode is set per-module below
    if not training:
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
    ret, _, _ = model(bd)
    handle.remove()
    loss = ret["loss"]
    model.zero_grad()
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    tag = "train" if training else "eval"
    keep = ["stem.0.kernel", "stem.3.kernel", "stage1.0.net.0.kernel", "stage2.1.net.0.kernel",
            "stage4.0.net.0.kernel", "up1.0.net.0.kernel", "up4.1.0.net.3.kernel", "up3.1.0.downsample.0.kernel",
            "classifier.0.weight", "stem.1.weight"]
    res = {f"{tag}_logits": captured["logits"].detach().numpy(), f"{tag}_loss": np.array(loss.item())}
    for k in keep:
        res[f"{tag}_grad/{k}"] = grads[k].numpy()
    res[f"{tag}_gradnorms"] = np.array([float(grads[n].norm()) for n, _ in model.named_parameters()])
    return cfg, model, coords_in, feats_in, labels, res


def gen_model(cls, name, in_dim, key, fname):
    out = {"backend": np.array(BACKEND_DESC)}
    for training in (True, False):
        cfg, model, coords, feats, labels, res = run_model(cls, name, in_dim, key, [21, 22], training)
        out.update(res)
    out.update(coords=coords, feats=feats, labels=labels)
    sd = model.state_dict()
    out["state_keys"] = np.array(list(sd.keys()))
    out["state_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    out["param_crc"] = np.array([zlib.crc32(v.numpy().tobytes()) for v in fill_parameters(cls(cfg, 20), seed=3)
                                 .state_dict().values()], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "N =", coords.shape[0], "loss(train/eval) =", out["train_loss"], out["eval_loss"])


# ----------------------------------------------------------------------------------------- TIAF (MinkUNetMsMm)
def make_tiaf_batch(seeds):
    """lidar_ms (voxelised fused cloud, 5 features) + camera stack + FOV cloud, collated like
    semantickitti_voxel_ms_mm.py does: SparseTensors through sparse_collate_fn, image stacks concatenated along
    the frame axis with cumulative `offset_img`."""
    samples, images, sems, frames = [], [], [], []
    for s in seeds:
        pts, lab = small_scan(s)
        pc_, inds, _ = dataset_voxelize(pts, lab)
        feat = np.concatenate([pts, np.ones_like(pts[:, :1])], 1)[inds]
        cam = synth_tiaf_sample(pc_[inds], feat, seed=s)
        samples.append({"lidar_ms": SparseTensor(feat, pc_[inds]), "targets_ms": SparseTensor(lab[inds], pc_[inds]),
                        "lidar_fov_ms": SparseTensor(cam["fov_feats"], cam["fov_coords"])})
        images.append(cam["images"])
        sems.append(cam["semantic"])
        frames.append(len(cam["images"]))
    batch = sparse_collate_fn(samples)
    batch["image_ms"] = torch.from_numpy(np.concatenate(images))
    batch["semantic_map_ms"] = torch.from_numpy(np.concatenate(sems))
    batch["offset_img"] = torch.tensor(np.cumsum(frames))
    batch["offset_ms"] = torch.tensor([0])
    return batch


def run_model_mm(cls, training):
    # cr must be 1.0: the FOV encoder's widths are fixed (unet3d.py:194), classifier_fusion expects 2 x 480 inputs
    cfg = make_model_cfg("MinkUNetMsMm", in_dim=5, cr=1.0, num_layer=[1] * 8, **TIAF_CFG)
    torch.manual_seed(0)
    model = cls(cfg, 20)
    fill_parameters(model, seed=3)
    bd = make_tiaf_batch([31, 32])
    for k in ("lidar_ms", "lidar_fov_ms"):
        bd[k].F = bd[k].F.float()
        bd[k].C = bd[k].C.int()
    inputs = {"coords": bd["lidar_ms"].C.numpy().copy(), "feats": bd["lidar_ms"].F.numpy().copy(),
              "labels": bd["targets_ms"].F.numpy().astype(np.int64),
              "fov_coords": bd["lidar_fov_ms"].C.numpy().copy(), "fov_feats": bd["lidar_fov_ms"].F.numpy().copy(),
              "images": bd["image_ms"].numpy().copy(), "semantic": bd["semantic_map_ms"].numpy().copy(),
              "offset_img": bd["offset_img"].numpy().copy()}
    captured = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, key=key: captured.__setitem__(key, o.detach().numpy().copy()))
             for key, m in (("logits", model.classifier), ("fusion_logits", model.classifier_fusion),
                            ("fov_logits", model.lidar_backbone.classifier),
                            ("image_logits", model.image_backbone.classifier))]
    model.train()
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()                      # Dropout2d(0.2) of the image branch is random: off for the fixture
        if not training and isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    ret, _, disp = model(bd)
    for h in hooks:
        h.remove()
    loss = ret["loss"]
    model.zero_grad()
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    tag = "train" if training else "eval"
    keep = ["stem.0.kernel", "stage4.1.net.0.kernel", "up4.1.0.net.3.kernel", "image_backbone.stem.0.conv1.weight",
            "image_backbone.stage3.conv2.weight", "image_backbone.up4.conv1.weight", "lidar_backbone.stem.0.kernel",
            "lidar_backbone.stage4.1.net.0.kernel", "lidar_backbone.classifier.0.weight",
            "classifier_fusion.0.weight", "classifier_fusion.3.weight"]
    res = {f"{tag}_{k}": v for k, v in captured.items()}
    res[f"{tag}_loss"] = np.array(loss.item())
    res[f"{tag}_loss_parts"] = np.array([float(disp[k]) for k in ("loss_lidar", "loss_fusion", "loss_image_s",
                                                                 "loss_image_d", "loss_image_lidar")])
    for k in keep:
        g = grads[k].numpy()
        res[f"{tag}_grad/{k}"] = g[..., ::4, ::4] if g.size > 200000 else g   # every 4th row / column: fixture size
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    res[f"{tag}_gradnames"] = np.array(names)
    res[f"{tag}_gradnorms"] = np.array([float(grads[n].norm()) for n in names])
    return cfg, model, inputs, res


def gen_model_mm(fname="model_minkunet_ms_mm.npz"):
    cls = _ref_env.setup_pcseg_mm()
    out = {"backend": np.array(BACKEND_DESC)}
    for training in (True, False):
        cfg, model, inputs, res = run_model_mm(cls, training)
        out.update(res)
    out.update(inputs)
    sd = model.state_dict()
    out["state_keys"] = np.array(list(sd.keys()))
    out["state_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "N =", inputs["coords"].shape[0], "N_fov =", inputs["fov_coords"].shape[0],
          "loss(train/eval) =", out["train_loss"], out["eval_loss"], out["train_loss_parts"])


# ----------------------------------------------------------------------------------------- KD (MinkUNetMsKd)
KD_CFG = dict(SAMPLING_TYPE="random", MAX_VOXEL=100000, FEAT_KD="mse", FEAT_KD_WEIGHT=10.0)
"""MODEL section of minkunet_mk34_cr10_fsa_kd.yaml:32-35; MAX_VOXEL raised so the random sub-sampling (:624-626) never
triggers and the fixture is deterministic."""


def make_kd_batch(seeds):
    """Teacher cloud = the voxelised scan; student cloud = 85 % of its voxels plus a few voxels the teacher lacks
    (the two aggregations of the reference differ in which history points they keep)."""
    samples = []
    for s in seeds:
        pts, lab = small_scan(s)
        pc_, inds, _ = dataset_voxelize(pts, lab)
        feat = np.concatenate([pts, np.ones_like(pts[:, :1])], 1)[inds]
        coords, labels = pc_[inds], lab[inds]
        rs = np.random.RandomState(s)
        keep = np.sort(rs.choice(len(coords), int(0.85 * len(coords)), replace=False))
        extra = coords[rs.choice(len(coords), 40, replace=False)] + np.array([[0, 0, 300]], dtype=coords.dtype)  # above every real voxel: no duplicates
        s_coords = np.concatenate([coords[keep], extra])
        s_feat = np.concatenate([feat[keep], feat[:40]])
        s_lab = np.concatenate([labels[keep], labels[:40]])
        order = np.lexsort((s_coords[:, 2], s_coords[:, 1], s_coords[:, 0]))
        samples.append({"lidar_ms": SparseTensor(s_feat[order], s_coords[order]),
                        "targets_ms": SparseTensor(s_lab[order], s_coords[order]),
                        "lidar_ms_gt": SparseTensor(feat, coords)})
    batch = sparse_collate_fn(samples)
    batch["offset_ms"] = torch.tensor([0])
    return batch


def run_model_kd(cls, training):
    cfg = make_model_cfg("MinkUNetMsKd", in_dim=5, cr=0.5, num_layer=[1] * 8, **KD_CFG)
    torch.manual_seed(0)
    model = cls(cfg, 20)
    fill_parameters(model, seed=3)
    bd = make_kd_batch([41, 42])
    for k in ("lidar_ms", "lidar_ms_gt"):
        bd[k].F = bd[k].F.float()
        bd[k].C = bd[k].C.int()
    inputs = {"coords": bd["lidar_ms"].C.numpy().copy(), "feats": bd["lidar_ms"].F.numpy().copy(),
              "labels": bd["targets_ms"].F.numpy().astype(np.int64),
              "gt_coords": bd["lidar_ms_gt"].C.numpy().copy(), "gt_feats": bd["lidar_ms_gt"].F.numpy().copy()}
    captured = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, key=key: captured.__setitem__(key, o.detach().numpy().copy()))
             for key, m in (("logits", model.classifier), ("teacher_logits", model.classifier_gt))]
    model.train()
    if not training:
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
    ret, _, disp = model(bd)
    for h in hooks:
        h.remove()
    loss = ret["loss"]
    model.zero_grad()
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    tag = "train" if training else "eval"
    res = {f"{tag}_{k}": v for k, v in captured.items()}
    res[f"{tag}_loss"] = np.array(loss.item())
    res[f"{tag}_loss_parts"] = np.array([float(disp["loss_seg"]), float(disp["loss_feat_kd"])])
    for k in ["stem.0.kernel", "stage2.1.net.0.kernel", "up4.1.0.net.3.kernel", "classifier.0.weight"]:
        res[f"{tag}_grad/{k}"] = grads[k].numpy()
    res[f"{tag}_teacher_has_grad"] = np.array([grads[n] is not None for n in grads if "_gt" in n])
    names = [n for n in grads if grads[n] is not None]
    res[f"{tag}_gradnames"] = np.array(names)
    res[f"{tag}_gradnorms"] = np.array([float(grads[n].norm()) for n in names])
    return cfg, model, inputs, res


def gen_model_kd(fname="model_minkunet_ms_kd.npz"):
    cls = _ref_env.setup_pcseg_kd()
    out = {"backend": np.array(BACKEND_DESC)}
    for training in (True, False):
        cfg, model, inputs, res = run_model_kd(cls, training)
        out.update(res)
    out.update(inputs)
    sd = model.state_dict()
    out["state_keys"] = np.array(list(sd.keys()))
    out["state_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "N =", inputs["coords"].shape[0], "N_gt =", inputs["gt_coords"].shape[0],
          "loss(train/eval) =", out["train_loss"], out["eval_loss"], out["train_loss_parts"])


# ----------------------------------------------------------------------------------------- multi-scan data stage
def gen_multiscan():
    SemMs, SemVoxMs, _ = _ref_env.setup_datasets()
    from pcseg.data.dataset.semantickitti.semantickitti_utils import LEARNING_MAP, LEARNING_MAP_INV
    T = 4
    out = {"steps": np.array(FLEXIBLE_STEPS_KITTI), "T": np.array(T)}
    inv = np.array([LEARNING_MAP_INV[i] for i in range(20)], dtype=np.uint32)
    out["learning_map_inv"] = inv.astype(np.int64)
    lm = np.zeros(260, dtype=np.int64)
    for k, v in LEARNING_MAP.items():
        lm[k] = v
    out["learning_map"] = lm
    samples = []
    for b, seed in enumerate([31, 32]):
        files, poses = {}, []
        for t in range(T + 1):                       # frame index t: current = T, history = T-1 .. 0
            pose = synth_pose(T - t)
            pts, lab = small_scan(1000 * seed + t, n=1500, pose=pose, scene_seed=seed)
            path = f"/data/sequences/00/velodyne/{t:06d}.bin"
            files[path] = pts
            files[path.replace("velodyne", "labels")[:-3] + "label"] = inv[lab].reshape(-1, 1)
            poses.append(pose)
            out[f"b{b}_points_t{t}"] = pts
            out[f"b{b}_rawlabels_t{t}"] = inv[lab]
            out[f"b{b}_pose_t{t}"] = pose
        ds = object.__new__(SemMs)
        ds.poses = {0: poses}
        ds.only_history, ds.split, ds.seq, ds.pseudo_mask, ds.trainval_seqs = True, "train", -1, "gt", ["00"]
        annos = [f"/data/sequences/00/velodyne/{t:06d}.bin" for t in range(T + 1)]
        real_fromfile = np.fromfile
        np.fromfile = lambda path, dtype=None, **kw: files[path].copy()
        try:
            raw_ms, ann_ms, mask_ms = ds.multiscan_fuse(annos, T, T, FLEXIBLE_STEPS_KITTI)
        finally:
            np.fromfile = real_fromfile
        raw = files[annos[T]]
        ann = np.vectorize(LEARNING_MAP.__getitem__)(files[annos[T].replace("velodyne", "labels")[:-3] + "label"] & 0xFFFF)
        fused = np.concatenate([raw, raw_ms[mask_ms]])                  # semantickitti_ms.py:143
        fused = ds.append_time_flag(raw, fused)                          # :144
        ann_fused = np.concatenate([ann, ann_ms[mask_ms]])               # :145
        out[f"b{b}_fused_all"] = raw_ms.astype(np.float32)
        out[f"b{b}_mask"] = mask_ms
        out[f"b{b}_raw_data_ms"] = fused.astype(np.float32)
        out[f"b{b}_labels_ms"] = ann_fused.reshape(-1).astype(np.int64)

        vox = object.__new__(SemVoxMs)
        vox.point_cloud_dataset = [{"xyzret": raw.copy(), "labels": ann.astype(np.uint8), "path": annos[T],
                                    "xyzret_ms": fused.astype(np.float32), "labels_ms": ann_fused.astype(np.uint8)}]
        vox.in_feature_dim, vox.training, vox.if_tta, vox.voxel_size, vox.num_points = 5, False, False, VOXEL, 3000000
        samples.append(vox.get_single_sample(0))
    batch = SemVoxMs.collate_batch(samples)
    for key in ("lidar", "lidar_ms", "inverse_map", "inverse_map_ms", "targets", "targets_ms"):
        out[f"batch_{key}_C"] = batch[key].C.numpy()
        out[f"batch_{key}_F"] = batch[key].F.numpy()
    for key in ("num_points", "num_points_ms", "offset", "offset_ms", "point_mask"):
        out[f"batch_{key}"] = batch[key].numpy()
    np.savez_compressed(os.path.join(HERE, "multiscan.npz"), **out)
    print("multiscan.npz: fused", [out[f"b{b}_raw_data_ms"].shape for b in range(2)], "voxels_ms",
          out["batch_lidar_ms_C"].shape)


# ----------------------------------------------------------------------------------------- mIoU parity
MIOU_SEEDS = list(range(5000, 5200))


def gen_miou(fname="miou_minkunet.npz"):
    """SURVEY 8(d) parity gate "mIoU parity": the reference MinkUNet (tiny, parameters fill_parameters(seed=3), BatchNorm
    on its running statistics) predicts 200 seeded synthetic scans one by one; stored are the per-voxel arg-max
    classes, the confusion matrix and the per-class IoU computed with the reference's own `fast_hist` / `per_class_iu`
    definitions (R/train.py:35-45, restated here on numpy: the trainer module itself is not importable)."""
    cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, num_layer=[1] * 8)
    torch.manual_seed(0)
    model = MinkUNet(cfg, 20)
    fill_parameters(model, seed=3)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    captured = {}

    def grab(_m, inputs, output):
        captured["logits"], captured["feat"] = output, inputs[0]

    model.classifier.register_forward_hook(grab)
    # seeded random weights put one class ahead everywhere (the mean feature dominates every logit): re-centre the
    # classifier on the mean point feature of the first scan and scale it up, so that the arg-max depends on what the
    # network computes per voxel.  The adjusted head travels in the fixture.
    batch, _ = make_batch([MIOU_SEEDS[0]])
    lidar = batch["lidar"]
    lidar.F, lidar.C = lidar.F.float(), lidar.C.int()
    with torch.no_grad():
        model({"lidar": lidar, "targets": batch["targets"], "offset": torch.tensor([0])})
        mu = captured["feat"].mean(0)
        w = model.classifier[0].weight
        w -= torch.outer(w @ mu, mu) / (mu @ mu)
        w *= 8.0 / float((captured["feat"] @ w.t()).std())
        model.classifier[0].bias.zero_()
    head_w, head_b = model.classifier[0].weight.detach().numpy().copy(), model.classifier[0].bias.detach().numpy().copy()
    preds, counts, margins = [], [], []
    hist = np.zeros((20, 20), dtype=np.int64)
    for s in MIOU_SEEDS:
        batch, raw = make_batch([s])
        lidar = batch["lidar"]
        lidar.F = lidar.F.float()
        lidar.C = lidar.C.int()
        with torch.no_grad():
            model({"lidar": lidar, "targets": batch["targets"], "offset": torch.tensor([0])})
        logits = captured["logits"].numpy()
        pred = logits.argmax(1)
        top2 = np.sort(logits, axis=1)[:, -2:]
        margins.append((top2[:, 1] - top2[:, 0]).astype(np.float32))
        lab = batch["targets"].F.numpy().astype(np.int64)
        k = (lab >= 0) & (lab < 20)
        hist += np.bincount(20 * lab[k] + pred[k], minlength=400).reshape(20, 20)
        preds.append(pred.astype(np.uint8))
        counts.append(len(pred))
    iou = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist) + 1e-9)
    np.savez_compressed(os.path.join(HERE, fname), backend=np.array(BACKEND_DESC), seeds=np.array(MIOU_SEEDS),
                        counts=np.array(counts), pred=np.concatenate(preds), hist=hist, iou=iou, head_weight=head_w,
                        head_bias=head_b, margin_quantiles=np.quantile(np.concatenate(margins), [0, 1e-4, 1e-3, 1e-2, 0.5]))
    print(fname, "scans", len(MIOU_SEEDS), "voxels", int(np.sum(counts)), "mIoU over present classes",
          float(iou[hist.sum(1) > 0].mean()))


if __name__ == "__main__":
    print("reference backend:", BACKEND_DESC)
    if "--only-miou" in sys.argv:
        gen_miou()
        sys.exit(0)
    if "--only-mm" in sys.argv:
        gen_model_mm()
        sys.exit(0)
    if "--only-kd" in sys.argv:
        gen_model_kd()
        sys.exit(0)
    gen_ops()
    gen_model(MinkUNet, "MinkUNet", 4, "lidar", "model_minkunet.npz")
    gen_model(MinkUNetMs, "MinkUNetMs", 5, "lidar_ms", "model_minkunet_ms.npz")
    gen_multiscan()
    gen_model_mm()
    gen_model_kd()
    gen_miou()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
