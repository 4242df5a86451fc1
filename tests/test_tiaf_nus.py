"""nuScenes TIAF dataset stage (nuscenes_ms_mm.py:196-401, nuscenes_voxel_ms_mm.py:77-262) against the fixture the REAL
reference's NuscenesMsMmDataset / NuscVoxelMsMmDataset produced (tests/golden/tiaf_nus.npz): the oracle's restatement and the
host-side keyframe selection on the CPU, the device stage (taseg_amd/data/nuscenes_tiaf.py: ts_fuse_sweeps, ts_project_cam,
voxelisation, collate) on the GPU box - bit for bit."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ts_oracle as O
from taseg_amd.data import nuscenes_tiaf as T
from taseg_amd.data.nuscenes import NuscSequence, select_sweeps, sweep_params

IMG_H, IMG_W = 68, 96


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(GOLDEN, "tiaf_nus.npz"), allow_pickle=False))


def _seq(g, b):
    p = f"b{b}_"
    return NuscSequence(is_key=g[p + "is_key"], key_index=g[p + "key_index"], timestamps=g[p + "timestamps"],
                        scene_tokens=g[p + "scene_tokens"].tolist(), local_indexes=g[p + "local_indexes"], s2l_r=g[p + "s2l_r"],
                        s2l_t=g[p + "s2l_t"], global_indexes=g[p + "global_indexes"], l2e_q=g[p + "key_l2e_q"],
                        l2e_t=g[p + "key_l2e_t"], e2g_q=g[p + "key_e2g_q"], e2g_t=g[p + "key_e2g_t"])


def _keys(g, b):
    p = f"b{b}_"
    return [dict(lidar2ego_rotation=g[p + "key_l2e_q"][i], lidar2ego_translation=g[p + "key_l2e_t"][i],
                 ego2global_rotation=g[p + "key_e2g_q"][i], ego2global_translation=g[p + "key_e2g_t"][i])
            for i in range(len(g[p + "key_l2e_q"]))]


def _cfg(g, b):
    p = f"b{b}_cfg_"
    return dict(multiscan_image=int(g[p + "multiscan_image"]), step_image=float(g[p + "step_image"]), interval=int(g[p + "interval"]),
                used_view=g[p + "used_view"].tolist(), paint_dist=float(g[p + "paint_dist"]), rng=int(g[p + "rng"]))


def _half(img_u8):
    """the reference's resize (:380): PIL bilinear to half size - host work, as file decoding is"""
    from PIL import Image
    im = Image.fromarray(img_u8)
    return np.asarray(im.resize((int(im.size[0] * 0.5), int(im.size[1] * 0.5)), Image.BILINEAR))


@pytest.mark.parametrize("b", [0, 1])
def test_oracle_restates_the_reference_image_side(g, b):
    p, cfg, keys = f"b{b}_", _cfg(g, b), _keys(g, b)
    index, lm = int(g[p + "index"]), g["learning_map"]
    gi = g[p + "global_indexes"]
    scene_of = [g[p + "scene_tokens"][i] for i in gi]
    stamps = [int(g[p + "timestamps"][i]) for i in gi]
    frames = O.nus_select_image_keyframes(keys, scene_of, index, cfg["multiscan_image"], cfg["step_image"], random.Random(cfg["rng"]))
    assert frames == g[p + "image_keyframes"].tolist()
    points = {i: g[f"{p}key{i}_points"] for i in range(len(keys))}
    labels = {i: lm[g[f"{p}key{i}_rawlabels"]] for i in range(len(keys))}
    height = int(g["height"])
    fov, fov_lab, images = [], [], []
    for batch_idx, d in enumerate(frames):
        i = index + d
        raw, ann = O.nus_tiaf_frame_cloud(keys, scene_of, stamps, index, d, frames[batch_idx - 1] if batch_idx else None,
                                          cfg["interval"], points, labels, cfg["paint_dist"])
        for view_idx, v in enumerate(cfg["used_view"]):
            cam = dict(lidar_cs_q=g[p + "lidar_cs_q"], lidar_cs_t=g[p + "lidar_cs_t"], lidar_pose_q=keys[i]["ego2global_rotation"],
                       lidar_pose_t=keys[i]["ego2global_translation"], cam_pose_q=g[f"{p}key{i}_view{v}_pose_q"],
                       cam_pose_t=g[f"{p}key{i}_view{v}_pose_t"], cam_cs_q=g[f"{p}view{v}_cs_q"], cam_cs_t=g[f"{p}view{v}_cs_t"],
                       intrinsic=g[f"{p}view{v}_intrinsic"])
            pts, pl, _ = O.nus_tiaf_fov_points(raw[:, :4].copy(), ann.copy(), cam, (IMG_H, IMG_W), height,
                                               batch_idx * len(cfg["used_view"]) + view_idx)
            fov.append(O.nus_transform_point(pts, keys[index], keys[i]))
            fov_lab.append(pl)
            img = _half(g[f"{p}key{i}_view{v}_image"]).astype(np.float32) / 255.
            images.append(img[..., [2, 1, 0]][2:])
    assert np.array_equal(np.concatenate(fov).astype(np.float32), g[p + "xyzret_fov_ms"])
    assert np.array_equal(np.concatenate(fov_lab).astype(np.int64), g[p + "labels_fov_ms"])
    assert np.array_equal(np.stack(images), g[p + "image_ms"])
    assert len(g[p + "xyzret_fov_ms"]) > 100


@pytest.mark.parametrize("b", [0, 1])
def test_host_side_keyframe_selection_and_camera_chain(g, b):
    p, cfg, seq = f"b{b}_", _cfg(g, b), _seq(g, b)
    index = int(g[p + "index"])
    assert T.select_image_keyframes(seq, index, cfg["multiscan_image"], cfg["step_image"], random.Random(cfg["rng"])) == \
        g[p + "image_keyframes"].tolist()
    assert T.select_image_keyframes(seq, index, 0, cfg["step_image"], random.Random(0)) == [0]
    v = cfg["used_view"][0]
    chain = T.camera_chain(g[p + "lidar_cs_q"], g[p + "lidar_cs_t"], seq.e2g_q[index], seq.e2g_t[index],
                           g[f"{p}key{index}_view{v}_pose_q"], g[f"{p}key{index}_view{v}_pose_t"], g[f"{p}view{v}_cs_q"],
                           g[f"{p}view{v}_cs_t"], g[f"{p}view{v}_intrinsic"])
    assert chain.shape == (57,) and np.array_equal(chain[48:], g[f"{p}view{v}_intrinsic"].reshape(-1))
    assert np.array_equal(chain[27:36].reshape(3, 3), O.quaternion_rotation_matrix(g[f"{p}key{index}_view{v}_pose_q"]).T)


def _device_sample(g, b):
    p, cfg, seq = f"b{b}_", _cfg(g, b), _seq(g, b)
    index, lm = int(g[p + "index"]), g["learning_map"]
    dev = "cuda"
    offsets = select_sweeps(seq, index, int(g["multiscan"]), float(g["step"]))
    assert offsets == g[p + "sample_list"].tolist()
    hp = [torch.from_numpy(g[f"{p}points_d{-d}"]).to(dev) for d in offsets]
    hs = [torch.from_numpy(g[f"{p}pseudo_d{-d}"].astype(np.int64)).to(dev) for d in offsets]
    hl = [torch.from_numpy(lm[g[f"{p}rawlabels_d{-d}"]] if f"{p}rawlabels_d{-d}" in g else np.zeros(len(g[f"{p}points_d{-d}"]), np.int64)).to(dev)
          for d in offsets]
    fsa = dict(points=torch.from_numpy(g[f"{p}key{index}_points"]).to(dev), labels=torch.from_numpy(lm[g[f"{p}key{index}_rawlabels"]]).to(dev),
               hist_points=hp, hist_labels=hl, hist_pseudo=hs, params=torch.from_numpy(sweep_params(seq, index, offsets)).to(dev))
    n_keys = len(seq.global_indexes)
    key_points = {i: torch.from_numpy(g[f"{p}key{i}_points"]).to(dev) for i in range(n_keys)}
    key_labels = {i: torch.from_numpy(lm[g[f"{p}key{i}_rawlabels"]]).to(dev) for i in range(n_keys)}
    view_cs = {v: (g[f"{p}view{v}_cs_q"], g[f"{p}view{v}_cs_t"], g[f"{p}view{v}_intrinsic"]) for v in cfg["used_view"]}
    cam_frames = {}
    for d in g[p + "image_keyframes"].tolist():
        for v in cfg["used_view"]:
            i = index + d
            cam_frames[(i, v)] = dict(pose_q=g[f"{p}key{i}_view{v}_pose_q"], pose_t=g[f"{p}key{i}_view{v}_pose_t"],
                                      image=torch.from_numpy(_half(g[f"{p}key{i}_view{v}_image"]).copy()).to(dev),
                                      semantic=torch.from_numpy(g[f"{p}key{i}_view{v}_semantic"]).to(dev))
    return T.build_nusc_tiaf_sample(fsa, seq, index, key_points, key_labels, (g[p + "lidar_cs_q"], g[p + "lidar_cs_t"]), view_cs,
                                    cam_frames, g["steps"].tolist(), cfg["multiscan_image"], cfg["step_image"], cfg["interval"],
                                    cfg["used_view"], cfg["paint_dist"], (IMG_W, IMG_H), 0.1, random.Random(cfg["rng"]), name=f"s{b}")


@pytest.mark.gpu
def test_device_stage_matches_the_reference(g):
    samples = [_device_sample(g, b) for b in range(2)]
    for b, s in enumerate(samples):
        p = f"b{b}_"
        assert s["_image_keyframes"] == g[p + "image_keyframes"].tolist()
        assert np.array_equal(s["_fov_points"].cpu().numpy(), g[p + "xyzret_fov_ms"])          # float32 bit for bit
        assert np.array_equal(s["_fov_labels"].cpu().numpy(), g[p + "labels_fov_ms"])
        assert np.array_equal(s["image_ms"].cpu().numpy(), g[p + "image_ms"])
        assert np.array_equal(s["semantic_map_ms"].cpu().numpy(), g[p + "semantic_map_ms"])
        assert list(s["depth_map_ms"].shape) == g[p + "depth_map_ms_shape"].tolist()
        assert list(s["lidar_map_ms"].shape) == g[p + "lidar_map_ms_shape"].tolist()
    batch = T.build_nusc_tiaf_batch(samples)
    for key in ("lidar", "lidar_ms", "inverse_map", "inverse_map_ms", "targets", "targets_ms", "targets_mapped", "targets_mapped_ms",
                "lidar_fov_ms", "targets_fov_ms"):
        assert np.array_equal(batch[key].C.cpu().numpy(), g[f"batch_{key}_C"]), key
        assert np.array_equal(batch[key].F.cpu().numpy(), g[f"batch_{key}_F"]), key
    for key in ("num_points", "num_points_ms", "offset", "offset_ms", "point_mask", "offset_img", "image_ms", "semantic_map_ms"):
        assert np.array_equal(batch[key].cpu().numpy().reshape(-1), g[f"batch_{key}"].reshape(-1)), key
    assert list(batch["depth_map_ms"].shape) == g["batch_depth_map_ms_shape"].tolist()
    assert list(batch["lidar_map_ms"].shape) == g["batch_lidar_map_ms_shape"].tolist()


@pytest.mark.gpu
def test_device_staged_batch_trains_the_nuscenes_tiaf_segmentor(g):
    """the stage's batch_dict is what MinkUNetMsMmNus consumes (nuscenes/minkunet_mk34_cr10_fsa_tiaf.yaml:38-55: 17 classes,
    IN_FEATURE_DIM 4, FOV losses on `targets_fov_ms`): one training step, five finite loss parts, gradients everywhere"""
    from taseg_amd.data.synthetic import TIAF_CFG, fill_parameters, make_model_cfg
    from taseg_amd.pcseg.model import build_network
    batch = T.build_nusc_tiaf_batch([_device_sample(g, b) for b in range(2)])
    cfg = make_model_cfg("MinkUNetMsMmNus", in_dim=4, cr=1.0, num_layer=[1] * 8, **TIAF_CFG)      # (the fusion head is cr 1.0 wide)
    model = fill_parameters(build_network(cfg, 17), seed=3).cuda().train()
    ret, tb, _ = model(batch)
    assert np.isfinite(float(tb["loss"])) and len([k for k in tb if k.startswith("loss")]) >= 5
    ret["loss"].backward()
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, missing[:5]
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters())
