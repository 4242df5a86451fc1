"""Golden vectors added in round 2 (build container only; reads /root/reference, inert on the GPU box).

    python tests/golden/make_golden_r2.py [--big] [--eval] [--ckpt] [--nus]      (no flag = all)

Like make_golden.py this runs the REAL reference (its torchsparse v1.4.0 Python + CPU kernels, its pcseg model and
dataset code; import arrangement in _ref_env.py) and stores inputs + outputs as fixtures; it contains no reference code.

  model_mk34_minkunet.npz / model_mk34_minkunet_ms.npz
      the BENCHMARKED configuration (BASELINE configs[1] / [2]): MinkUNet / MinkUNetMs mk34, cr 1.0, bs 2, two 90 degree
      sectors of full-resolution 64-beam synthetic scans (>= 20k voxels each, ~6.5 rulebook pairs per voxel at stride 1
      like the bench workload): logits (every 8th row), loss, the norm of every parameter gradient, strided samples of
      24 gradients and of BatchNorm running statistics - train-mode and running-statistics BatchNorm.  Stored twice:
      `ref32_*` from the reference (fp32) and `oracle64_*` from our oracle evaluated in float64 (the yardstick that
      separates summation-order noise of ANY fp32 implementation from real disagreement).
  eval_minkunet.npz / eval_minkunet_ms.npz
      the reference's eval branch (minkunet.py:435-455, minkunet_ms.py:433-458) on dataset-collated batches: per-scan
      point_predict / point_predict_logits / point_labels through inverse_map (+ point_mask), return_logit form, and a
      3-vote test-time-augmentation batch with the vote sum the trainer forms (R/train.py:474-477).
  ckpt_minkunet_ms_ref.pth
      a checkpoint in the reference's on-disk format (R/train.py:319-342: epoch / it / model_state / optimizer_state /
      scaler_state / scheduler_state; keys `module.`-prefixed as a DDP-wrapped model saves them) of the reference's own
      MinkUNetMs class, with the logits that model produces (ckpt_minkunet_ms_ref.npz).
  multiscan_nus.npz
      nuScenes FSA stage (nuscenes_ms.py:226-373, nuscenes_voxel_ms.py): sweep selection by driven distance, ego-box
      filter, transform_point, time delta, class-step mask, voxelisation + collate.
"""
import os
import sys
import time
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import _ref_env  # noqa: E402
from taseg_amd.data.synthetic import (fill_parameters, make_model_cfg, strided_sample, synth_pose,  # noqa: E402
                                      synth_scan)

torch.set_num_threads(1)
BACKEND_DESC = _ref_env.setup_torchsparse()
import torchsparse  # noqa: E402,F401  (the reference library)
from torchsparse import SparseTensor  # noqa: E402
from torchsparse.utils.collate import sparse_collate_fn  # noqa: E402
from torchsparse.utils.quantize import sparse_quantize  # noqa: E402

MinkUNet, MinkUNetMs = _ref_env.setup_pcseg()
VOXEL = 0.05

BIG_GRADS = ["stem.0.kernel", "stem.3.kernel", "stem.1.weight", "stem.1.bias", "stage1.0.net.0.kernel",
             "stage1.1.net.0.kernel", "stage2.1.net.0.kernel", "stage2.1.downsample.0.kernel",
             "stage2.1.downsample.1.weight", "stage3.2.net.3.kernel", "stage4.0.net.0.kernel", "stage4.3.net.0.kernel",
             "stage4.6.net.4.weight", "up1.0.net.0.kernel", "up1.1.0.net.0.kernel", "up1.1.0.downsample.0.kernel",
             "up2.1.1.net.3.kernel", "up3.0.net.0.kernel", "up3.1.0.net.0.kernel", "up4.0.net.0.kernel",
             "up4.1.0.net.0.kernel", "up4.1.1.net.3.kernel", "classifier.0.weight", "classifier.0.bias"]
BIG_STATS = ["stem.1.running_mean", "stem.1.running_var", "stage4.6.net.4.running_mean", "stage4.6.net.4.running_var",
             "up4.1.1.net.4.running_mean", "up4.1.1.net.4.running_var"]


def sector_scan(seed, half_width_deg=45.0):
    """a 90 degree azimuth sector of a full-resolution (64 beams x 2083 azimuth steps, 120k points) synthetic scan"""
    pts, lab = synth_scan(seed, n_points=120000)
    az = np.degrees(np.arctan2(pts[:, 1], pts[:, 0]))
    keep = np.abs(az) <= half_width_deg
    return pts[keep], lab[keep]


def dataset_voxelize(points):
    pc_ = np.round(points[:, :3] / VOXEL).astype(np.int32)          # semantickitti_voxel.py:119-127
    pc_ -= pc_.min(0, keepdims=1)
    _, inds, inverse = sparse_quantize(pc_, return_index=True, return_inverse=True)
    return pc_, inds, inverse


def big_batch(seeds, in_dim):
    samples = []
    for s in seeds:
        pts, lab = sector_scan(s)
        pc_, inds, _ = dataset_voxelize(pts)
        feat = pts if in_dim == 4 else np.concatenate([pts, np.ones_like(pts[:, :1])], 1)
        samples.append({"lidar": SparseTensor(feat[inds], pc_[inds]), "targets": SparseTensor(lab[inds], pc_[inds])})
    return sparse_collate_fn(samples)


def summarise(prefix, logits, loss, grads, stats):
    out = {prefix + "logits": logits[::8].astype(np.float32), prefix + "loss": np.float64(loss)}
    out[prefix + "gradnorms"] = np.array([float(np.linalg.norm(g.astype(np.float64))) for g in grads.values()])
    for k in BIG_GRADS:
        out[prefix + "grad/" + k] = strided_sample(grads[k], 2048).astype(np.float32)
    for k in BIG_STATS:
        out[prefix + "stat/" + k] = strided_sample(stats[k], 2048).astype(np.float32)
    return out


def run_big_reference(cls, cfg, key, batch, training):
    torch.manual_seed(0)
    model = fill_parameters(cls(cfg, 20), seed=3)
    model.train()
    if not training:
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
    lidar = SparseTensor(batch["lidar"].F.float().clone(), batch["lidar"].C.int().clone())
    bd = {key: lidar, ("targets" if key == "lidar" else "targets_ms"): batch["targets"],
          ("offset" if key == "lidar" else "offset_ms"): torch.tensor([0])}
    captured = {}
    h = model.classifier.register_forward_hook(lambda m, i, o: captured.__setitem__("logits", o))
    ret, _, _ = model(bd)
    h.remove()
    model.zero_grad()
    ret["loss"].backward()
    grads = {n: p.grad.numpy() for n, p in model.named_parameters()}
    stats = {n: b.detach().numpy() for n, b in model.named_buffers()}
    return captured["logits"].detach().numpy(), float(ret["loss"]), grads, stats, [n for n, _ in model.named_parameters()]


def run_big_oracle64(cfg, name, batch, training):
    from oracle import model as OM
    from taseg_amd.pcseg.model import build_network
    model = fill_parameters(build_network(cfg, 20), seed=3)
    learn = [n for n, _ in model.named_parameters()]
    params = {k: v.detach().double().clone().requires_grad_(k in learn) for k, v in model.state_dict().items()
              if v.is_floating_point()}
    om = OM.OracleMinkUNet(params, cfg, backend="numpy", training=training)
    coords = batch["lidar"].C.int().numpy()
    feats = batch["lidar"].F.double()
    labels = batch["targets"].F.long()
    fwd = om.forward_minkunet if name == "MinkUNet" else om.forward_minkunet_ms
    logits = fwd(coords, feats)
    loss = OM.loss_ce_lovasz(logits, labels)
    loss.backward()
    grads = {n: params[n].grad.numpy() for n in learn}
    stats = {n: params[n].detach().numpy() for n in params if "running" in n}
    return logits.detach().numpy(), float(loss), grads, stats


def gen_big(cls, name, in_dim, key, fname, seeds=(61, 62)):
    cfg = make_model_cfg(name, in_dim=in_dim, cr=1.0)                    # mk34: NUM_LAYER [2,3,4,6,2,2,2,2]
    batch = big_batch(seeds, in_dim)
    out = {"backend": np.array(BACKEND_DESC), "seeds": np.array(seeds), "coords": batch["lidar"].C.int().numpy(),
           "feats": batch["lidar"].F.float().numpy(), "labels": batch["targets"].F.numpy().astype(np.int64)}
    for training in (True, False):
        tag = "train_" if training else "eval_"
        t0 = time.time()
        logits, loss, grads, stats, names = run_big_reference(cls, cfg, key, batch, training)
        out.update(summarise("ref32_" + tag, logits, loss, grads, stats))
        t1 = time.time()
        logits64, loss64, grads64, stats64 = run_big_oracle64(cfg, name, batch, training)
        out.update(summarise("oracle64_" + tag, logits64, loss64, {n: grads64[n] for n in names}, stats64))
        rel = max(np.linalg.norm(grads[n] - grads64[n]) / max(np.linalg.norm(grads64[n]), 1e-30) for n in names)
        print(f"{fname} {tag}: N = {len(out['coords'])}, reference {t1 - t0:.0f} s, oracle64 {time.time() - t1:.0f} s, "
              f"loss {loss:.6f} / {loss64:.6f}, max |logit diff| {np.abs(logits - logits64).max():.2e}, "
              f"worst relative gradient error of the reference vs fp64 {rel:.2e}", flush=True)
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB")


# ----------------------------------------------------------------------------------------- eval branch + TTA votes
FLEX_KITTI = [0, 0, 2, 2, 2, 2, 2, 2, 2, 0, 4, 4, 4, 0, 4, 0, 2, 4, 2, 2]      # minkunet_mk34_cr10_fsa.yaml:17


def small_scan(seed, n=1500, **kw):
    return synth_scan(seed, n_points=n, n_beams=16, n_az=360, **kw)


def kitti_ms_dataset(seeds, T=4):
    """The reference's multi-scan voxel dataset (semantickitti_voxel_ms.py) over synthetic sequences held in memory:
    returns the dataset object (one entry per seed; file IO of the frame reader replaced by array lookups)."""
    SemMs, SemVoxMs, _ = _ref_env.setup_datasets()
    from pcseg.data.dataset.semantickitti.semantickitti_utils import LEARNING_MAP, LEARNING_MAP_INV
    inv = np.array([LEARNING_MAP_INV[i] for i in range(20)], dtype=np.uint32)
    entries = []
    for seed in seeds:
        files, poses = {}, []
        for t in range(T + 1):
            pose = synth_pose(T - t)
            pts, lab = small_scan(1000 * seed + t, pose=pose, scene_seed=seed)
            path = f"/data/sequences/00/velodyne/{t:06d}.bin"
            files[path] = pts
            files[path.replace("velodyne", "labels")[:-3] + "label"] = inv[lab].reshape(-1, 1)
            poses.append(pose)
        ds = object.__new__(SemMs)
        ds.poses = {0: poses}
        ds.only_history, ds.split, ds.seq, ds.pseudo_mask, ds.trainval_seqs = True, "train", -1, "gt", ["00"]
        annos = [f"/data/sequences/00/velodyne/{t:06d}.bin" for t in range(T + 1)]
        real_fromfile = np.fromfile
        np.fromfile = lambda path, dtype=None, **kw: files[path].copy()
        try:
            raw_ms, ann_ms, mask_ms = ds.multiscan_fuse(annos, T, T, FLEX_KITTI)
        finally:
            np.fromfile = real_fromfile
        raw = files[annos[T]]
        ann = np.vectorize(LEARNING_MAP.__getitem__)(files[annos[T].replace("velodyne", "labels")[:-3] + "label"] & 0xFFFF)
        fused = ds.append_time_flag(raw, np.concatenate([raw, raw_ms[mask_ms]]))
        ann_fused = np.concatenate([ann, ann_ms[mask_ms]])
        entries.append({"xyzret": raw.copy(), "labels": ann.astype(np.uint8), "path": f"/data/sequences/00/velodyne/{seed:06d}.bin",
                        "xyzret_ms": fused.astype(np.float32), "labels_ms": ann_fused.astype(np.uint8)})
    class Frames(list):
        """the frame reader hands out fresh arrays on every access (it reads files); the voxel dataset augments in place"""

        def __getitem__(self, i):
            return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in list.__getitem__(self, i).items()}

    vox = object.__new__(SemVoxMs)
    vox.point_cloud_dataset = Frames(entries)
    vox.in_feature_dim, vox.training, vox.if_tta, vox.voxel_size, vox.num_points = 5, False, False, VOXEL, 3000000
    vox.if_flip = vox.if_scale = vox.if_jitter = vox.if_rotate = False
    vox.scale_axis, vox.scale_range = "xyz", [0.95, 1.05]
    return vox, SemVoxMs


BATCH_SPARSE = ("lidar", "lidar_ms", "inverse_map", "inverse_map_ms", "targets", "targets_ms", "targets_mapped",
                "targets_mapped_ms")
BATCH_DENSE = ("num_points", "num_points_ms", "offset", "offset_ms", "point_mask")


def dump_batch(prefix, batch):
    out = {}
    for key in BATCH_SPARSE:
        out[f"{prefix}{key}_C"] = batch[key].C.numpy()
        out[f"{prefix}{key}_F"] = batch[key].F.numpy()
    for key in BATCH_DENSE:
        out[f"{prefix}{key}"] = batch[key].numpy()
    out[prefix + "name"] = np.array(batch["name"])
    return out


def clone_batch(batch):
    out = {}
    for k, v in batch.items():
        if isinstance(v, SparseTensor):
            out[k] = SparseTensor(v.F.clone(), v.C.clone())
        elif isinstance(v, torch.Tensor):
            out[k] = v.clone()
        else:
            out[k] = list(v)
    for k in ("lidar", "lidar_ms"):
        out[k].F, out[k].C = out[k].F.float(), out[k].C.int()
    return out


def dump_eval(prefix, ret):
    out = {}
    for b in range(len(ret["point_predict"])):
        out[f"{prefix}point_predict_{b}"] = np.asarray(ret["point_predict"][b])
        out[f"{prefix}point_labels_{b}"] = np.asarray(ret["point_labels"][b])
        out[f"{prefix}point_predict_logits_{b}"] = np.asarray(ret["point_predict_logits"][b])
    out[prefix + "name"] = np.array(ret["name"])
    return out


def gen_eval(fname="eval_ms.npz", votes=3):
    vox, SemVoxMs = kitti_ms_dataset([51, 52])
    out = {"backend": np.array(BACKEND_DESC), "votes": np.array(votes)}
    batch = SemVoxMs.collate_batch([vox.get_single_sample(0), vox.get_single_sample(1)])
    out.update(dump_batch("batch_", batch))
    # test-time augmentation: `votes` rotated / scaled copies of scan 0, one batch entry per vote
    # (semantickitti_voxel_ms.py:66-72,103-121 + collate_batch_tta; the scale factor comes from numpy's global RNG)
    vox.if_tta, vox.votes_min, vox.votes_max = True, 0, votes
    np.random.seed(7)
    tta = SemVoxMs.collate_batch_tta([vox[0]])
    out.update(dump_batch("tta_", tta))
    models = {}
    for name, cls, in_dim, key in (("minkunet", MinkUNet, 4, "lidar"), ("minkunet_ms", MinkUNetMs, 5, "lidar_ms")):
        cfg = make_model_cfg("MinkUNet" if in_dim == 4 else "MinkUNetMs", in_dim=in_dim, cr=0.5, num_layer=[1] * 8)
        torch.manual_seed(0)
        model = fill_parameters(cls(cfg, 20), seed=3).eval()
        models[name] = model
        with torch.no_grad():
            out.update(dump_eval(f"{name}_", model(clone_batch(batch))))
            ret = model(clone_batch(tta))
        out.update(dump_eval(f"{name}_tta_", ret))
        # the trainer's vote accumulation (R/train.py:474-477, 505-508): sum of the per-vote point logits, arg-max,
        # uint32 .label payload
        acc = np.asarray(ret["point_predict_logits"][0]).copy()
        for count in range(1, votes):
            acc += ret["point_predict_logits"][count]
        out[f"{name}_tta_sum"] = acc
        out[f"{name}_tta_label"] = np.expand_dims(np.argmax(acc, axis=1), axis=1).astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB; points per scan",
          [len(out[f"minkunet_ms_point_predict_{b}"]) for b in range(2)], "votes", votes)


# ----------------------------------------------------------------------------------------- checkpoint interop
def gen_ckpt(fname="ckpt_minkunet_ms_ref"):
    """A checkpoint as the reference trainer writes it (R/train.py:319-342) from the reference's own MinkUNetMs, keys
    `module.`-prefixed like the state_dict of a DistributedDataParallel-wrapped model (what
    BaseSegmentor.load_params strips, base_segmentors.py:16-26), + the logits this model gives on a small batch."""
    cfg = make_model_cfg("MinkUNetMs", in_dim=5, cr=0.125, num_layer=[1] * 8)
    torch.manual_seed(0)
    model = fill_parameters(MinkUNetMs(cfg, 20), seed=11)
    opt = torch.optim.SGD(model.parameters(), lr=0.24, momentum=0.9, weight_decay=1e-4, nesterov=True)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0)
    state = {"epoch": 3, "it": 1234,
             "model_state": type(model.state_dict())(("module." + k, v.cpu()) for k, v in model.state_dict().items()),
             "optimizer_state": opt.state_dict(), "scaler_state": torch.cuda.amp.GradScaler(enabled=False).state_dict(),
             "scheduler_state": {k: v for k, v in sched.state_dict().items() if k != "lr_lambdas"}}
    torch.save(state, os.path.join(HERE, fname + ".pth"))
    vox, SemVoxMs = kitti_ms_dataset([53])
    batch = SemVoxMs.collate_batch([vox.get_single_sample(0)])
    model.eval()
    with torch.no_grad():
        ret = model(clone_batch(batch))
    out = {"backend": np.array(BACKEND_DESC)}
    out.update(dump_batch("batch_", batch))
    out.update(dump_eval("ref_", ret))
    out["n_entries"] = np.array(len(state["model_state"]))
    np.savez_compressed(os.path.join(HERE, fname + ".npz"), **out)
    print(fname, os.path.getsize(os.path.join(HERE, fname + ".pth")) // 1024, "KiB checkpoint,",
          sum(p.numel() for p in model.parameters()), "parameters")


# ----------------------------------------------------------------------------------------- nuScenes FSA stage
NUS_RAW_OF_CLASS = [0, 9, 14, 16, 17, 18, 21, 2, 12, 22, 23, 24, 25, 26, 27, 28, 30]   # one raw lidarseg id per class
FLEX_NUS = [0, 1, 1, 1, 3, 1, 1, 3, 1, 3, 3, 0, 1, 1, 1, 1, 1]      # nuscenes/minkunet_mk34_cr10_fsa.yaml:22


def _install_pyquaternion():
    """pyquaternion (a dependency of nuscenes-devkit; unpinned by the reference, R/docs/INSTALL.md:24-27) is not in this
    image.  nuscenes_ms.py:348-373 only needs `Quaternion(q).rotation_matrix`; the published algorithm (pyquaternion
    0.9.9: normalise, then the lower-right 3x3 of Q(q) . Qbar(q)^T) is restated ONCE, in oracle/ts_oracle.py, and served
    to the reference under the library's import name.  Recorded in the fixture as `quaternion`."""
    import types
    from oracle import ts_oracle as O

    class Quaternion:
        def __init__(self, q):
            self.q = np.asarray(q, dtype=np.float64)

        @property
        def rotation_matrix(self):
            return O.quaternion_rotation_matrix(self.q)

    mod = types.ModuleType("pyquaternion")
    mod.Quaternion = Quaternion
    sys.modules["pyquaternion"] = mod


def _yaw_quat(yaw, pitch=0.0):
    cy, sy, cp, sp = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2)
    return [float(cy * cp), float(-sy * sp), float(cy * sp), float(sy * cp)]          # w, x, y, z


def nus_scene(seed, n_pts=700):
    """Two synthetic scenes in nuScenes' bookkeeping (mmdet3d-style info dicts as nuscenes_ms.py reads them): keyframes
    every 6th frame, 5 sweeps between them, the car driving 0.42 m and turning 0.6 degrees per frame.  Returns the
    attributes of a NuscenesMsDataset (infos, sweep list, index tables), the in-memory files and the fake devkit tables."""
    from oracle import ts_oracle as O
    from taseg_amd.data.synthetic import KITTI_TO_NUSC
    l2e_q, l2e_t = [0.70710678, 0.0, 0.0, -0.70710678], [0.94, 0.0, 0.0]
    files, infos, sweeps, scene_tokens, local_indexes, global_indexes = {}, [], [], [], [], {}
    sample_tab, lidarseg_tab = {}, {}
    frame = 0
    for scene, n_frames in (("sceneA", 7), ("sceneB", 25)):
        first = frame
        for f in range(n_frames):
            g = first + f
            yaw = np.deg2rad(0.6 * g)
            e2g_q, e2g_t = _yaw_quat(yaw, np.deg2rad(0.2)), [0.42 * g, 0.05 * np.sin(0.3 * g), 0.0]
            world = np.eye(4)
            l2e_r, e2g_r = O.quaternion_rotation_matrix(l2e_q), O.quaternion_rotation_matrix(e2g_q)
            world[:3, :3] = e2g_r @ l2e_r
            world[:3, 3] = e2g_r @ np.array(l2e_t) + np.array(e2g_t)
            pts, lab = synth_scan(100 * seed + g, n_points=n_pts, n_beams=16, n_az=360, pose=world.astype(np.float32),
                                  scene_seed=seed)
            rs = np.random.RandomState(7000 + g)
            ego = rs.choice(len(pts), 25, replace=False)                  # a few returns from the ego vehicle itself
            pts[ego, 0] = rs.uniform(-1.2, 1.2, 25).astype(np.float32)
            pts[ego, 1] = rs.uniform(-1.8, 1.8, 25).astype(np.float32)
            raw5 = np.concatenate([pts, np.zeros((len(pts), 1), np.float32)], 1)
            cls = np.asarray(KITTI_TO_NUSC, dtype=np.uint8)[lab]
            is_key = (f % 6 == 0)
            stamp = 1_600_000_000_000_000 + 50_000 * g
            if is_key:
                token, sd = f"sample{g:03d}", f"sd{g:03d}"
                path = f"./data/nuscenes/samples/LIDAR_TOP/{g:03d}.bin"
                info = {"lidar_path": path, "token": token, "lidar2ego_rotation": l2e_q, "lidar2ego_translation": l2e_t,
                        "ego2global_rotation": e2g_q, "ego2global_translation": e2g_t, "timestamp": stamp}
                infos.append(info)
                global_indexes[len(infos) - 1] = g
                sample_tab[token] = {"data": {"LIDAR_TOP": sd}, "scene_token": scene}
                lidarseg_tab[sd] = {"filename": f"lidarseg/{sd}_lidarseg.bin"}
                files["/nus/" + path[16:]] = raw5
                files["/nus/" + lidarseg_tab[sd]["filename"]] = np.asarray(NUS_RAW_OF_CLASS, dtype=np.uint8)[cls]
                sweeps.append(info)
                pseudo_token = sd
            else:
                sd = f"sweep{g:03d}"
                path = f"./data/nuscenes/sweeps/LIDAR_TOP/{g:03d}.bin"
                sweeps.append({"data_path": path, "sample_data_token": sd, "timestamp": stamp, "_pose": (e2g_q, e2g_t)})
                files["/nus/" + path[16:]] = raw5
                pseudo_token = sd
            files["/YourHome/PCSeg/logs/voxel/nuscenes/minkunet_mk34_cr10/default/results/lidarseg/trainval_sweep_notta/"
                  + pseudo_token + "_lidarseg.bin"] = cls.copy()
            scene_tokens.append(scene)
            frame += 1
    # every frame's "father" = the next keyframe at or after it (the keyframe whose info lists it as a sweep); its
    # sensor2lidar_* maps the sweep's lidar frame into the father's (mmdet3d obtain_sensor2top)
    key_pos = sorted(global_indexes.items(), key=lambda kv: kv[1])
    for g, sw in enumerate(sweeps):
        father = next(i for i, gp in key_pos if gp >= g and scene_tokens[gp] == scene_tokens[g])
        local_indexes.append(father)
        if "data_path" in sw:
            e2g_q, e2g_t = sw.pop("_pose")
            fi = infos[father]
            l2e_r = O.quaternion_rotation_matrix(l2e_q)
            e2g_r_s, e2g_r = O.quaternion_rotation_matrix(e2g_q), O.quaternion_rotation_matrix(fi["ego2global_rotation"])
            R = (l2e_r.T @ e2g_r_s.T) @ (np.linalg.inv(e2g_r).T @ np.linalg.inv(l2e_r).T)
            T = (np.array(l2e_t) @ e2g_r_s.T + np.array(e2g_t)) @ (np.linalg.inv(e2g_r).T @ np.linalg.inv(l2e_r).T)
            T -= np.array(fi["ego2global_translation"]) @ (np.linalg.inv(e2g_r).T @ np.linalg.inv(l2e_r).T) \
                + np.array(l2e_t) @ np.linalg.inv(l2e_r).T
            sw["sensor2lidar_rotation"], sw["sensor2lidar_translation"] = R.T, T
    gi = [global_indexes[i] for i in range(len(infos))]
    return dict(infos=infos, sweeps=sweeps, scene_tokens=scene_tokens, local_indexes=local_indexes, global_indexes=gi,
                files=files, sample_tab=sample_tab, lidarseg_tab=lidarseg_tab)


def gen_nus(fname="multiscan_nus.npz", multiscan=4, step=1.0):
    import yaml
    _install_pyquaternion()
    _ref_env._pkg("pcseg.data.dataset.nuscenes", os.path.join(_ref_env.REF, "pcseg", "data", "dataset", "nuscenes"))
    for name in ("np.float", "np.bool"):
        if not hasattr(np, name.split(".")[1]):
            setattr(np, name.split(".")[1], float if name.endswith("float") else bool)   # aliases numpy >= 1.24 dropped
    from pcseg.data.dataset.nuscenes.nuscenes_ms import NuscenesMsDataset
    from pcseg.data.dataset.nuscenes.nuscenes_voxel_ms import NuscVoxelMsDataset
    with open(os.path.join(_ref_env.REF, "pcseg", "data", "dataset", "nuscenes", "nuscenes.yaml")) as f:
        learning_map = yaml.safe_load(f)["learning_map"]
    out = {"backend": np.array(BACKEND_DESC), "quaternion": np.array("pyquaternion absent: rotation_matrix restated "
                                                                     "(oracle.ts_oracle.quaternion_rotation_matrix)"),
           "multiscan": np.array(multiscan), "step": np.array(step), "steps": np.array(FLEX_NUS)}
    lm = np.zeros(256, dtype=np.int64)
    for k, v in learning_map.items():
        lm[k] = v
    out["learning_map"] = lm
    samples = []
    for b, seed in enumerate((71, 72)):
        sc = nus_scene(seed)

        class FakeNusc:
            dataroot = "/nus"

            def get(self, table, token):
                return (sc["sample_tab"] if table == "sample" else sc["lidarseg_tab"])[token]

        ds = object.__new__(NuscenesMsDataset)
        ds.root_path, ds.data_path_ceph, ds.split, ds.seq, ds.augment, ds.tta = "/nus", None, "val", -1, "none", False
        ds.nusc, ds.nusc_infos, ds.nusc_infos_sweep = FakeNusc(), sc["infos"], sc["sweeps"]
        ds.global_indexes, ds.local_indexes, ds.scene_tokens = sc["global_indexes"], sc["local_indexes"], sc["scene_tokens"]
        ds.token2samplelist, ds.learning_map = {}, learning_map
        ds.multiscan, ds.step, ds.flexible_steps, ds.pseudo_mask = multiscan, step, FLEX_NUS, "mink_sweep_notta"
        real_fromfile = np.fromfile
        np.fromfile = lambda path, dtype=None, count=-1, **kw: sc["files"][path].copy()
        index = len(sc["infos"]) - 1 - b              # last / second-to-last keyframe of scene B
        try:
            pc_data = ds[index]
            lidar_sd = sc["sample_tab"][sc["infos"][index]["token"]]["data"]["LIDAR_TOP"]
            sample_list = list(ds.token2samplelist[lidar_sd])
            raw_ms, ann_ms, pseudo_ms, mask_ms = ds.multiscan_fuse(index, lidar_sd, multiscan, step, FLEX_NUS)
        finally:
            np.fromfile = real_fromfile
        out[f"b{b}_index"] = np.array(index)
        out[f"b{b}_sample_list"] = np.array(sample_list)
        out[f"b{b}_fused_all"] = raw_ms.astype(np.float32)
        out[f"b{b}_labels_all"] = ann_ms.reshape(-1).astype(np.int64)
        out[f"b{b}_pseudo_all"] = pseudo_ms.reshape(-1).astype(np.int64)
        out[f"b{b}_mask"] = mask_ms
        out[f"b{b}_xyzret"] = pc_data["xyzret"].astype(np.float32)
        out[f"b{b}_xyzret_ms"] = pc_data["xyzret_ms"].astype(np.float32)
        out[f"b{b}_labels"] = pc_data["labels"].reshape(-1).astype(np.int64)
        out[f"b{b}_labels_ms"] = pc_data["labels_ms"].reshape(-1).astype(np.int64)
        # the scene itself: what the oracle / the device stage start from
        g0 = sc["global_indexes"][index]
        out[f"b{b}_global_index"] = np.array(g0)
        out[f"b{b}_scene_tokens"] = np.array(sc["scene_tokens"])
        out[f"b{b}_local_indexes"] = np.array(sc["local_indexes"])
        out[f"b{b}_global_indexes"] = np.array(sc["global_indexes"])
        out[f"b{b}_is_key"] = np.array(["lidar_path" in s for s in sc["sweeps"]])
        out[f"b{b}_timestamps"] = np.array([s["timestamp"] for s in sc["sweeps"]], dtype=np.int64)
        key_of = {id(info): i for i, info in enumerate(sc["infos"])}
        out[f"b{b}_key_index"] = np.array([key_of.get(id(s), -1) for s in sc["sweeps"]])
        out[f"b{b}_key_l2e_q"] = np.array([i["lidar2ego_rotation"] for i in sc["infos"]], dtype=np.float64)
        out[f"b{b}_key_l2e_t"] = np.array([i["lidar2ego_translation"] for i in sc["infos"]], dtype=np.float64)
        out[f"b{b}_key_e2g_q"] = np.array([i["ego2global_rotation"] for i in sc["infos"]], dtype=np.float64)
        out[f"b{b}_key_e2g_t"] = np.array([i["ego2global_translation"] for i in sc["infos"]], dtype=np.float64)
        s2l_r = np.zeros((len(sc["sweeps"]), 3, 3))
        s2l_t = np.zeros((len(sc["sweeps"]), 3))
        for g, s in enumerate(sc["sweeps"]):
            if "data_path" in s:
                s2l_r[g], s2l_t[g] = s["sensor2lidar_rotation"], s["sensor2lidar_translation"]
        out[f"b{b}_s2l_r"], out[f"b{b}_s2l_t"] = s2l_r, s2l_t
        for d in sorted(set(sample_list)):
            s = sc["sweeps"][g0 + d]
            path = "/nus/" + (s["lidar_path"] if "lidar_path" in s else s["data_path"])[16:]
            tok = sc["sample_tab"][s["token"]]["data"]["LIDAR_TOP"] if "lidar_path" in s else s["sample_data_token"]
            out[f"b{b}_points_d{-d}"] = sc["files"][path]
            out[f"b{b}_pseudo_d{-d}"] = sc["files"]["/YourHome/PCSeg/logs/voxel/nuscenes/minkunet_mk34_cr10/default/results/"
                                                    "lidarseg/trainval_sweep_notta/" + tok + "_lidarseg.bin"]
            if "lidar_path" in s:
                out[f"b{b}_rawlabels_d{-d}"] = sc["files"]["/nus/" + sc["lidarseg_tab"][tok]["filename"]]
        out[f"b{b}_points_cur"] = sc["files"]["/nus/" + sc["infos"][index]["lidar_path"][16:]]
        out[f"b{b}_rawlabels_cur"] = sc["files"]["/nus/" + sc["lidarseg_tab"][lidar_sd]["filename"]]

        vox = object.__new__(NuscVoxelMsDataset)
        vox.point_cloud_dataset = [pc_data]
        vox.in_feature_dim, vox.training, vox.if_tta, vox.voxel_size, vox.num_points = 4, False, False, 0.1, 1000000
        samples.append(vox.get_single_sample(0))
    batch = NuscVoxelMsDataset.collate_batch(samples)
    out.update(dump_batch("batch_", batch))
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB; sample lists",
          [out[f"b{b}_sample_list"].tolist() for b in range(2)], "fused", [out[f"b{b}_xyzret_ms"].shape for b in range(2)],
          "voxels_ms", out["batch_lidar_ms_C"].shape)


# ----------------------------------------------------------------------------------------- TIAF dataset stage
def gen_tiaf_data(fname="tiaf_data.npz", T=8, multiscan=4, multiscan_image=8, step_image=4, height=60, width=192):
    """The reference's TIAF dataset path on two synthetic sequences held in memory: SemantickittiMsMmDataset.__getitem__
    (multiscan_fuse with camera frames, get_fov_points, ring ids; semantickitti_ms_mm.py:143-461) ->
    SemkittiVoxelMsMmDataset.get_single_sample / collate_batch (semantickitti_voxel_ms_mm.py:79-266).  Camera images and
    semantic maps are random arrays served through patched Image.open / np.load; the image is a little larger than the
    (height, width) crop on one axis and smaller on the other, so that crop mask and zero padding both act."""
    import types
    from PIL import Image
    sys.modules.setdefault("mmcv", types.ModuleType("mmcv"))
    for alias, typ in (("int", int), ("bool", bool), ("float", float)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    _ref_env.setup_datasets()
    from pcseg.data.dataset.semantickitti.semantickitti_ms_mm import SemantickittiMsMmDataset
    from pcseg.data.dataset.semantickitti.semantickitti_voxel_ms_mm import SemkittiVoxelMsMmDataset
    from pcseg.data.dataset.semantickitti.semantickitti_utils import LEARNING_MAP_INV
    inv = np.array([LEARNING_MAP_INV[i] for i in range(20)], dtype=np.uint32)
    img_h, img_w = 64, 180                       # rows: 64 > HEIGHT 60 (cropped); columns: 180 < WIDTH 192 (zero padded)
    # KITTI-style calibration: camera looks along +x of the lidar (x_cam = -y, y_cam = -z, z_cam = x), pinhole P2
    tr = np.array([[0.0, -1.0, 0.0, 0.02], [0.0, 0.0, -1.0, -0.07], [1.0, 0.0, 0.0, -0.27], [0.0, 0.0, 0.0, 1.0]])
    p2 = np.array([[95.0, 0.0, 90.5, 4.5], [0.0, 95.0, 30.25, 0.2], [0.0, 0.0, 1.0, 0.003]])
    proj = np.matmul(p2, tr)
    out = {"backend": np.array(BACKEND_DESC), "T": np.array(T), "multiscan": np.array(multiscan),
           "multiscan_image": np.array(multiscan_image), "step_image": np.array(step_image), "height": np.array(height),
           "width": np.array(width), "proj": proj, "steps": np.array(FLEX_KITTI),
           "learning_map_inv": inv.astype(np.int64)}
    samples = []
    for b, seed in enumerate([81, 82]):
        files, poses = {}, []
        rs = np.random.RandomState(900 + seed)
        for t in range(T + 1):
            pose = synth_pose(T - t)
            pts, lab = small_scan(1000 * seed + t, pose=pose, scene_seed=seed)
            raw = inv[lab].copy()
            if t < T:
                raw[:150] = 254                   # moving persons in the history: class 6 by learning_map, never aggregated
            path = f"/data/sequences/00/velodyne/{t:06d}.bin"
            files[path] = pts
            files[path.replace("velodyne", "labels")[:-3] + "label"] = raw.reshape(-1, 1)
            poses.append(pose)
            out[f"b{b}_points_t{t}"] = pts
            out[f"b{b}_rawlabels_t{t}"] = raw
            out[f"b{b}_pose_t{t}"] = pose
            if (T - t) % step_image == 0:
                img = rs.randint(0, 256, size=(img_h, img_w, 3)).astype(np.uint8)
                sem = rs.randint(0, 20, size=(img_h, img_w, 1)).astype(np.float32)
                files[path.replace("velodyne", "image_2").replace(".bin", ".png")] = img
                files[path.replace("velodyne", "semantic_map_dilate").replace(".bin", ".npy")] = sem
                out[f"b{b}_image_t{t}"] = img
                out[f"b{b}_semantic_t{t}"] = sem
        ds = object.__new__(SemantickittiMsMmDataset)
        ds.poses, ds.proj_matrix = {0: poses}, {0: proj}
        ds.only_history, ds.split, ds.seq, ds.pseudo_mask, ds.trainval_seqs = True, "val", -1, "gt", ["00"]
        ds.if_scribble, ds.augment, ds.dynamic_step, ds.fov_dist = False, "none", False, -1
        ds.multiscan, ds.flexible_steps, ds.multiscan_image, ds.step_image = multiscan, FLEX_KITTI, multiscan_image, step_image
        ds.height, ds.width, ds.image_jitter, ds.image_flip, ds.flip_ratio = height, width, False, False, 0.5
        ds.annos = [f"/data/sequences/00/velodyne/{t:06d}.bin" for t in range(T + 1)]
        ds.annos_another = list(ds.annos)
        real_fromfile, real_load, real_open, real_array = np.fromfile, np.load, Image.open, np.array
        np.fromfile = lambda path, dtype=None, **kw: files[path].copy()
        np.load = lambda path, *a, **kw: files[path].copy()
        Image.open = lambda path, *a, **kw: Image.fromarray(files[path])
        # the reference is written against numpy 1.x, where np.array(..., copy=False) means "copy only if needed" (:428)
        np.array = lambda obj, *a, copy=True, **kw: real_array(obj, *a, copy=(None if copy is False else copy), **kw)
        try:
            pc_data = ds[T]
        finally:
            np.fromfile, np.load, Image.open, np.array = real_fromfile, real_load, real_open, real_array
        for k in ("xyzret", "xyzret_ms", "xyzret_fov_ms", "image_ms", "semantic_map_ms"):
            out[f"b{b}_{k}"] = np.asarray(pc_data[k], dtype=np.float32) if k != "image_ms" else pc_data[k][..., ::1][:, ::3, ::3].copy()
        out[f"b{b}_image_ms_shape"] = np.array(pc_data["image_ms"].shape)
        out[f"b{b}_labels"] = pc_data["labels"].reshape(-1).astype(np.int64)
        out[f"b{b}_labels_ms"] = pc_data["labels_ms"].reshape(-1).astype(np.int64)
        vox = object.__new__(SemkittiVoxelMsMmDataset)
        vox.point_cloud_dataset = [pc_data]
        vox.in_feature_dim, vox.training, vox.if_tta, vox.voxel_size, vox.num_points = 5, False, False, VOXEL, 3000000
        vox.eval_range = [0, 1000]
        samples.append(vox.get_single_sample(0))
    batch = SemkittiVoxelMsMmDataset.collate_batch(samples)
    for key in BATCH_SPARSE + ("lidar_fov_ms",):
        out[f"batch_{key}_C"] = batch[key].C.numpy()
        out[f"batch_{key}_F"] = batch[key].F.numpy()
    for key in BATCH_DENSE + ("offset_img",):
        out[f"batch_{key}"] = batch[key].numpy()
    out["batch_image_ms_sub"] = batch["image_ms"].numpy()[:, :, ::3, ::3].copy()      # NCHW, every 3rd pixel
    out["batch_image_ms_shape"] = np.array(batch["image_ms"].shape)
    out["batch_semantic_map_ms"] = batch["semantic_map_ms"].numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB; fov points", [out[f"b{b}_xyzret_fov_ms"].shape for b in range(2)],
          "fov voxels", out["batch_lidar_fov_ms_C"].shape, "images", out["batch_image_ms_shape"].tolist())


# ----------------------------------------------------------------------------------------- dense-rulebook op fixtures
def gen_ops_dense(fname="ops_dense.npz"):
    """Op-level convolution fixtures from the reference on a cloud with a DENSE rulebook (the round-1 ops.npz cloud has 1.08
    pairs per voxel at stride 1): a 45 degree sector of a full-resolution scan, ~6.5 pairs per voxel - submanifold k3 with
    full-tile channel counts (32 -> 64: the split-bf16 MFMA kernels) and a ragged one (4 -> 20), the strided k2 convolution
    and its transposed mirror, forward + both gradients through the reference's autograd Function; rulebooks included."""
    import torchsparse.nn.functional as F
    from torchsparse.nn.utils import get_kernel_offsets
    pts, lab = sector_scan(63, half_width_deg=15.0)
    pc_, inds, _ = dataset_voxelize(pts)
    coords = torch.from_numpy(np.concatenate([pc_[inds], np.zeros((len(inds), 1), np.int32)], 1)).int()
    n = coords.shape[0]
    out = {"backend": np.array(BACKEND_DESC), "coords": coords.numpy()}
    refs = F.sphash(coords)
    res = F.sphashquery(F.sphash(coords, get_kernel_offsets(3, 1, 1)), refs)
    out["k3_nbsizes"] = torch.sum(res != -1, dim=1).numpy().astype(np.int32)
    nb = torch.nonzero(res != -1)
    nb[:, 0] = res.view(-1)[nb[:, 0] * res.size(1) + nb[:, 1]]
    out["k3_nbmaps"] = nb.numpy().astype(np.int32)
    # inputs are NOT stored: tests regenerate them from the same seeded torch CPU generator, in the same order
    # (tests/conftest.py::dense_ops_inputs); outputs with one row per voxel are stored for every 4th voxel
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import dense_ops_inputs
    inp = dense_ops_inputs(n)
    for tag in ("full", "ragged"):
        xin, w, gy = (inp[f"{tag}_{k}"].clone() for k in ("x", "w", "gy"))
        xin.requires_grad_()
        w.requires_grad_()
        y = F.conv3d(SparseTensor(xin, coords, 1), w, 3)
        y.F.backward(gy)
        out.update({f"{tag}_y": y.F.detach().numpy()[::4], f"{tag}_gx": xin.grad.numpy()[::4], f"{tag}_gw": w.grad.numpy()})
    xin, wd, wu = inp["t_x"].clone().requires_grad_(), inp["t_wd"].clone().requires_grad_(), inp["t_wu"].clone().requires_grad_()
    st = SparseTensor(xin, coords, 1)
    st.cmaps[st.stride] = coords
    yd = F.conv3d(st, wd, 2, stride=2)
    yu = F.conv3d(yd, wu, 2, stride=2, transposed=True)
    yu.F.backward(inp["t_gy"])
    out.update({"t_coords_d": yd.C.numpy(), "t_yd": yd.F.detach().numpy()[::2], "t_yu": yu.F.detach().numpy()[::4],
                "t_gx": xin.grad.numpy()[::4], "t_gwd": wd.grad.numpy(), "t_gwu": wu.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB;", n, "voxels,", len(out["k3_nbmaps"]), "pairs =",
          round(len(out["k3_nbmaps"]) / n, 2), "per voxel")


# ----------------------------------------------------------------------------------------- training-step trajectory
def gen_train_steps(fname="train_steps_minkunet_ms.npz", steps=4):
    """The reference's training step (R/train.py:398-416: zero_grad, forward, backward, clip_grad_norm_(10),
    torch.optim.SGD(momentum 0.9, weight decay 1e-4) as R/pcseg/optim/__init__.py:15-21 builds it) repeated over
    alternating batches with the reference's MinkUNetMs: losses per step, gradient norm before clipping, strided samples
    of the parameters after the last step and of BatchNorm running statistics."""
    from torch.nn.utils import clip_grad_norm_
    cfg = make_model_cfg("MinkUNetMs", in_dim=5, cr=0.5, num_layer=[1] * 8)
    torch.manual_seed(0)
    model = fill_parameters(MinkUNetMs(cfg, 20), seed=5)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.02, weight_decay=1e-4, momentum=0.9)
    batches = []
    for seeds in ((91, 92), (93, 94)):
        samples = []
        for sd in seeds:
            pts, lab = small_scan(sd, n=3000)
            pc_, inds, _ = dataset_voxelize(pts)
            feat = np.concatenate([pts, np.ones_like(pts[:, :1])], 1)
            samples.append({"lidar": SparseTensor(feat[inds], pc_[inds]), "targets": SparseTensor(lab[inds], pc_[inds])})
        batches.append(sparse_collate_fn(samples))
    out = {"backend": np.array(BACKEND_DESC), "steps": np.array(steps), "lr": np.array(0.02), "momentum": np.array(0.9),
           "weight_decay": np.array(1e-4), "max_norm": np.array(10.0)}
    for i, bt in enumerate(batches):
        out[f"coords{i}"] = bt["lidar"].C.int().numpy()
        out[f"feats{i}"] = bt["lidar"].F.float().numpy()
        out[f"labels{i}"] = bt["targets"].F.numpy().astype(np.int64)
    losses, norms = [], []
    for it in range(steps):
        bt = batches[it % 2]
        opt.zero_grad()
        bd = {"lidar_ms": SparseTensor(bt["lidar"].F.float().clone(), bt["lidar"].C.int().clone()), "targets_ms": bt["targets"],
              "offset_ms": torch.tensor([0])}
        ret, _, _ = model(bd)
        loss = ret["loss"].mean()
        loss.backward()
        norms.append(float(clip_grad_norm_(model.parameters(), 10.0)))
        opt.step()
        losses.append(float(loss))
    out["losses"], out["grad_norms"] = np.array(losses), np.array(norms)
    names = [n for n, _ in model.named_parameters()]
    out["param_names"] = np.array(names)
    out["param_norms"] = np.array([float(p.detach().double().norm()) for _, p in model.named_parameters()])
    for n, p in model.named_parameters():
        if n in ("stem.0.kernel", "stage2.1.net.0.kernel", "stage4.1.net.3.kernel", "up2.1.0.net.0.kernel",
                 "up4.1.0.net.3.kernel", "classifier.0.weight", "stem.1.weight", "up1.0.net.1.bias"):
            out["param/" + n] = strided_sample(p.detach().numpy(), 2048)
    for n, bbuf in model.named_buffers():
        if n in ("stem.1.running_mean", "stem.1.running_var", "stage4.1.net.4.running_var", "up4.1.0.net.4.running_mean"):
            out["stat/" + n] = bbuf.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB; losses", [round(l, 5) for l in losses],
          "grad norms", [round(n, 3) for n in norms])


if __name__ == "__main__":
    print("reference backend:", BACKEND_DESC)
    args = set(sys.argv[1:])
    every = not args
    if every or "--big" in args:
        gen_big(MinkUNet, "MinkUNet", 4, "lidar", "model_mk34_minkunet.npz")
        gen_big(MinkUNetMs, "MinkUNetMs", 5, "lidar_ms", "model_mk34_minkunet_ms.npz")
    if every or "--eval" in args:
        gen_eval()
    if every or "--ckpt" in args:
        gen_ckpt()
    if every or "--nus" in args:
        gen_nus()
    if every or "--tiaf" in args:
        gen_tiaf_data()
    if every or "--ops-dense" in args:
        gen_ops_dense()
    if every or "--train-steps" in args:
        gen_train_steps()
