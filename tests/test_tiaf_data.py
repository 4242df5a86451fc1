"""TIAF dataset stage (camera side + three-cloud voxelisation) against what the REAL reference's dataset code produced
(tests/golden/tiaf_data.npz: SemantickittiMsMmDataset.__getitem__ -> SemkittiVoxelMsMmDataset.get_single_sample ->
collate_batch on two synthetic sequences): the oracle's restatement on the CPU, the device stage on the GPU box."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ts_oracle as O


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(GOLDEN, "tiaf_data.npz"), allow_pickle=False))


def _frames(g, b):
    T, step = int(g["T"]), int(g["step_image"])
    return T, step, [t for t in range(T + 1) if (T - t) % step == 0]


def test_oracle_camera_side_matches_reference_golden(g):
    height, width = int(g["height"]), int(g["width"])
    for b in range(2):
        T, step, with_img = _frames(g, b)
        fov, stack = [], []
        for t in sorted(with_img, reverse=True):                      # newest first (semantickitti_ms_mm.py:378-382)
            img = g[f"b{b}_image_t{t}"]
            raw_fov, keep = O.tiaf_fov_points(g[f"b{b}_points_t{t}"], g["proj"], (img.shape[1], img.shape[0]), height, width,
                                              (T - t) // step)
            assert 0 < keep.sum() < len(keep)
            if t != T:
                raw_fov = O.fuse_scan(raw_fov, g[f"b{b}_pose_t{T}"], g[f"b{b}_pose_t{t}"])
            fov.append(raw_fov)
            stack.append(O.tiaf_crop_image(img, height, width))
        assert np.array_equal(np.concatenate(fov).astype(np.float32), g[f"b{b}_xyzret_fov_ms"])      # bit for bit
        assert np.array_equal(np.stack(stack)[:, ::3, ::3], g[f"b{b}_image_ms"])
        assert list(np.stack(stack).shape) == g[f"b{b}_image_ms_shape"].tolist()
        # pixel rows carry the frame offset, columns stay inside the crop; a row >= HEIGHT was cut, columns are padded
        rows, cols = g[f"b{b}_xyzret_fov_ms"][:, 4], g[f"b{b}_xyzret_fov_ms"][:, 5]
        assert rows.max() < height * len(with_img) and cols.max() < width and (rows % height).max() <= height - 1
        # ring id column of the single-frame cloud, time flag column of the fused one
        assert np.array_equal(O.kitti_ring_id(g[f"b{b}_points_t{T}"]).astype(np.float32), g[f"b{b}_xyzret"][:, 4])
        n_cur = len(g[f"b{b}_points_t{T}"])
        assert g[f"b{b}_xyzret_ms"][:n_cur, 4].min() == 1 and g[f"b{b}_xyzret_ms"][n_cur:, 4].max() == 0
        # the moving-person labels (raw 254) of the history scans were not aggregated
        inv = g["learning_map_inv"]
        kept = []
        for t in range(T - int(g["multiscan"]), T):
            kept.append(O.history_mask(g[f"b{b}_rawlabels_t{t}"], t - T, g["steps"].tolist(), inv))
            assert not kept[-1][:150].any()
        assert n_cur + sum(int(k.sum()) for k in kept) == len(g[f"b{b}_xyzret_ms"])


@pytest.mark.gpu
def test_device_tiaf_stage_matches_reference_golden(g):
    from taseg_amd.data import semantickitti as SK
    from taseg_amd.data.tiaf import build_tiaf_batch, build_tiaf_sample
    height, width = int(g["height"]), int(g["width"])
    proj = torch.from_numpy(g["proj"]).cuda()
    samples = []
    for b in range(2):
        T, step, with_img = _frames(g, b)
        frames = {}
        for t in range(T + 1):
            raw = g[f"b{b}_rawlabels_t{t}"].astype(np.int64)
            f = {"points": torch.from_numpy(g[f"b{b}_points_t{t}"]).cuda(), "labels": torch.from_numpy(SK._LUT[raw]).cuda(),
                 "pseudo": torch.from_numpy(SK._CANON[raw]).cuda(), "pose": torch.from_numpy(g[f"b{b}_pose_t{t}"]).cuda()}
            if t in with_img:
                f["image"] = torch.from_numpy(g[f"b{b}_image_t{t}"]).cuda()
                f["semantic"] = torch.from_numpy(g[f"b{b}_semantic_t{t}"]).cuda()
            frames[t - T] = f
        samples.append(build_tiaf_sample(frames, g["steps"].tolist(), int(g["multiscan"]), step, proj, (height, width), 0.05,
                                         name=f"/data/sequences/00/velodyne/{T:06d}.bin"))
    batch = build_tiaf_batch(samples)
    for key in ("lidar", "lidar_ms", "lidar_fov_ms", "inverse_map", "inverse_map_ms", "targets", "targets_ms", "targets_mapped",
                "targets_mapped_ms"):
        assert np.array_equal(batch[key].C.cpu().numpy(), g[f"batch_{key}_C"]), key
        got = batch[key].F.cpu().numpy()
        assert np.array_equal(got.astype(g[f"batch_{key}_F"].dtype), g[f"batch_{key}_F"]), key
    for key in ("num_points", "num_points_ms", "offset", "offset_ms", "point_mask", "offset_img"):
        assert np.array_equal(batch[key].cpu().numpy().reshape(-1), g[f"batch_{key}"].reshape(-1)), key
    img = batch["image_ms"].cpu().numpy()
    assert list(img.shape) == g["batch_image_ms_shape"].tolist() and np.array_equal(img[:, :, ::3, ::3], g["batch_image_ms_sub"])
    assert np.array_equal(batch["semantic_map_ms"].cpu().numpy(), g["batch_semantic_map_ms"])
