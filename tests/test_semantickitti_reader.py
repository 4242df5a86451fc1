"""SemanticKITTI reader (taseg_amd/data/semantickitti.py): label tables and pose parsing on the CPU, and - on the GPU box -
a SemanticKITTI directory tree written from the golden fixture's scans, read back and pushed through the device stage,
bit for bit against what the REAL reference's dataset code produced (tests/golden/multiscan.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import ts_oracle as O
from taseg_amd.data import semantickitti as SK


def _write_tree(root, seq, points, raw_labels, poses, tr=None):
    d = os.path.join(root, str(seq).zfill(2))
    os.makedirs(os.path.join(d, "velodyne"), exist_ok=True)
    os.makedirs(os.path.join(d, "labels"), exist_ok=True)
    tr = np.eye(4) if tr is None else tr
    with open(os.path.join(d, "calib.txt"), "w") as f:
        for key in ("P0", "P1", "P2", "P3"):
            f.write(key + ": " + " ".join(repr(float(v)) for v in np.eye(4)[:3].reshape(-1)) + "\n")
        f.write("Tr: " + " ".join(repr(float(v)) for v in tr[:3].reshape(-1)) + "\n")
    tr_inv = np.linalg.inv(tr)
    with open(os.path.join(d, "poses.txt"), "w") as f:
        for p in poses:            # camera-frame pose whose velodyne-frame form Tr^-1 P Tr is `p`
            cam = tr @ np.asarray(p, dtype=np.float64) @ tr_inv
            f.write(" ".join(repr(float(v)) for v in cam[:3].reshape(-1)) + "\n")
    for t, (pts, lab) in enumerate(zip(points, raw_labels)):
        np.asarray(pts, dtype=np.float32).tofile(os.path.join(d, "velodyne", f"{t:06d}.bin"))
        np.asarray(lab, dtype=np.uint32).tofile(os.path.join(d, "labels", f"{t:06d}.label"))
    return d


def test_label_tables_match_reference(g_multiscan):
    lm = g_multiscan["learning_map"]
    inv = g_multiscan["learning_map_inv"]
    assert all(int(lm[k]) == v for k, v in SK.LEARNING_MAP.items()) and int(lm.sum()) == sum(SK.LEARNING_MAP.values())
    assert [SK.LEARNING_MAP_INV[c] for c in range(20)] == inv.tolist()
    # canonical-id table: class c only for the raw id LEARNING_MAP_INV[c]; moving-object ids map to a class but are not canonical
    assert SK._CANON[30] == 6 and SK._CANON[254] == -1 and SK._LUT[254] == 6 and SK._CANON[0] == 0


def test_pose_parsing_round_trip(tmp_path):
    rs = np.random.RandomState(0)
    tr = np.eye(4)
    tr[:3, :3] = np.linalg.qr(rs.randn(3, 3))[0]
    tr[:3, 3] = rs.randn(3)
    poses = []
    for t in range(5):
        p = np.eye(4)
        a = 0.01 * t
        p[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        p[:3, 3] = [1.1 * t, 0.02 * t, 0]
        poses.append(p)
    _write_tree(str(tmp_path), 3, [np.zeros((2, 4), np.float32)] * 5, [np.zeros(2, np.uint32)] * 5, poses, tr)
    seq = SK.KittiSequence(str(tmp_path), 3)
    assert len(seq) == 5 and seq.poses[0].dtype == np.float32
    for got, want in zip(seq.poses, poses):
        assert np.allclose(got, want, atol=1e-6)
    calib = SK.parse_calibration(os.path.join(str(tmp_path), "03", "calib.txt"))
    assert set(calib) == {"P0", "P1", "P2", "P3", "Tr"} and np.allclose(calib["Tr"], tr)
    assert seq.points(1).shape == (2, 4) and seq.raw_labels(1).tolist() == [0, 0]


@pytest.mark.gpu
def test_files_to_device_stage_matches_reference_golden(g_multiscan, tmp_path):
    from taseg_amd.data.stage import build_multiscan_batch
    g = g_multiscan
    Tn = int(g["T"])
    steps = g["steps"].tolist()
    samples = []
    for b in range(2):
        _write_tree(str(tmp_path), b, [g[f"b{b}_points_t{t}"] for t in range(Tn + 1)],
                    [g[f"b{b}_rawlabels_t{t}"] for t in range(Tn + 1)], [g[f"b{b}_pose_t{t}"] for t in range(Tn + 1)])
        seq = SK.KittiSequence(str(tmp_path), b)
        for t in range(Tn + 1):
            assert np.array_equal(seq.poses[t], g[f"b{b}_pose_t{t}"])
        samples.append(SK.multiscan_sample(seq, Tn, Tn, steps))
        assert samples[-1]["deltas"] == list(range(-Tn, 0))
    batch = build_multiscan_batch(samples, 0.05, steps)
    for key in ("lidar", "lidar_ms", "inverse_map", "inverse_map_ms", "targets", "targets_ms"):
        assert np.array_equal(batch[key].C.cpu().numpy(), g[f"batch_{key}_C"]), key
        assert np.array_equal(batch[key].F.cpu().numpy().astype(g[f"batch_{key}_F"].dtype), g[f"batch_{key}_F"]), key
    for key in ("num_points", "num_points_ms", "offset", "offset_ms", "point_mask"):
        assert np.array_equal(batch[key].cpu().numpy().reshape(-1), g[f"batch_{key}"].reshape(-1)), key


@pytest.mark.gpu
def test_moving_object_ids_are_not_aggregated(g_multiscan, tmp_path):
    """the reference compares the RAW pseudo label with LEARNING_MAP_INV[class] (semantickitti_ms.py:303-308): a moving
    person (raw 254 -> class 6, step 2) in a history scan is dropped, a static one (raw 30) is kept; first frames of a
    sequence have a shorter history"""
    from taseg_amd.data.stage import _fuse_history
    g = g_multiscan
    Tn = int(g["T"])
    steps = g["steps"].tolist()
    pts = [g[f"b0_points_t{t}"] for t in range(Tn + 1)]
    raw = [g[f"b0_rawlabels_t{t}"].astype(np.uint32).copy() for t in range(Tn + 1)]
    for t in range(Tn):
        raw[t][:200] = 254
        raw[t][200:400] = 30
    _write_tree(str(tmp_path), 0, pts, raw, [g[f"b0_pose_t{t}"] for t in range(Tn + 1)])
    seq = SK.KittiSequence(str(tmp_path), 0)
    s = SK.multiscan_sample(seq, Tn, Tn, steps)
    t = len(s["points"]) - 1
    raw_all, lab_all, keep = _fuse_history(s["points"][t], s["labels"][t], s["points"][:t], s["labels"][:t], s["poses"][t],
                                           s["poses"][:t], s["deltas"], steps, s["pseudo"])
    keep = keep.cpu().numpy()
    inv = g["learning_map_inv"]
    want, fused = [np.ones(len(pts[Tn]), dtype=bool)], [np.concatenate([pts[Tn]], 0)]
    for i, d in enumerate(s["deltas"]):
        want.append(O.history_mask(raw[i], d, steps, inv))
        fused.append(O.fuse_scan(pts[i], g[f"b0_pose_t{Tn}"], g[f"b0_pose_t{i}"]))
    want = np.concatenate(want)
    assert np.array_equal(keep, want)
    assert np.array_equal(raw_all[:, :4].cpu().numpy(), np.concatenate(fused))
    n0 = len(pts[Tn])
    first = slice(n0 + len(pts[0]), n0 + len(pts[0]) + 400)          # history scan delta = -3: step-2 classes are skipped
    second = slice(n0 + len(pts[0]) + len(pts[1]), n0 + len(pts[0]) + len(pts[1]) + 400)      # delta = -2
    assert not keep[first].any() and not keep[second][:200].any() and keep[second][200:400].all()
    short = SK.multiscan_sample(seq, 1, Tn, steps)
    assert short["deltas"] == [-1] and len(short["points"]) == 2
