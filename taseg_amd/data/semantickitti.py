"""SemanticKITTI reader for the device data stage: the file side of
R/pcseg/data/dataset/semantickitti/semantickitti_ms.py:120-149, 263-400 - velodyne .bin scans, .label annotations,
calib.txt / poses.txt - feeding `taseg_amd.data.stage.build_multiscan_batch`, which does the pose fuse, the class-step
filter and the voxelisation on the GPU.

    <root>/<seq>/velodyne/000000.bin   float32 [n, 4]  x, y, z, remission
    <root>/<seq>/labels/000000.label   uint32  [n]     (instance << 16) | semantic id
    <root>/<seq>/calib.txt, poses.txt  KITTI odometry text files (camera poses; "Tr" = camera <- velodyne)

    seq = KittiSequence("/data/SemanticKITTI/sequences", 8)
    sample = multiscan_sample(seq, frame=120, multiscan=16, steps=FLEXIBLE_STEPS_KITTI, device="cuda")
    batch = build_multiscan_batch([sample], 0.05, FLEXIBLE_STEPS_KITTI)

Host work per sample = reading ~17 files and one 4x4 product per pose; everything per point runs on the device.
Augmentation (LaserMix / PolarMix / rotate-scale-flip) is outside the scope contract.
"""
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

__all__ = ["LEARNING_MAP", "LEARNING_MAP_INV", "parse_calibration", "parse_poses", "KittiSequence", "multiscan_sample"]

# semantic-kitti label maps (the dataset's published label definition, semantic-kitti.yaml `learning_map` /
# `learning_map_inv`; the reference keeps a copy in semantickitti_utils.py): raw id -> training class, class -> canonical raw id
LEARNING_MAP = {0: 0, 1: 0, 10: 1, 11: 2, 13: 5, 15: 3, 16: 5, 18: 4, 20: 5, 30: 6, 31: 7, 32: 8, 40: 9, 44: 10, 48: 11,
                49: 12, 50: 13, 51: 14, 52: 0, 60: 9, 70: 15, 71: 16, 72: 17, 80: 18, 81: 19, 99: 0, 252: 1, 253: 7, 254: 6,
                255: 8, 256: 5, 257: 5, 258: 4, 259: 5}
LEARNING_MAP_INV = {0: 0, 1: 10, 2: 11, 3: 15, 4: 18, 5: 20, 6: 30, 7: 31, 8: 32, 9: 40, 10: 44, 11: 48, 12: 49, 13: 50,
                    14: 51, 15: 70, 16: 71, 17: 72, 18: 80, 19: 81}

_LUT = np.zeros(260, dtype=np.int64)
for _k, _v in LEARNING_MAP.items():
    _LUT[_k] = _v
_CANON = np.full(260, -1, dtype=np.int64)            # raw id -> class whose canonical id it is (else -1)
for _c, _raw in LEARNING_MAP_INV.items():
    _CANON[_raw] = _c


def _rows_to_pose(values) -> np.ndarray:
    pose = np.zeros((4, 4))
    pose[0, 0:4], pose[1, 0:4], pose[2, 0:4] = values[0:4], values[4:8], values[8:12]
    pose[3, 3] = 1.0
    return pose


def parse_calibration(filename: str) -> Dict[str, np.ndarray]:
    """calib.txt -> {key: 4x4} (semantickitti_ms.py:349-375)."""
    calib = {}
    with open(filename) as f:
        for line in f:
            if ":" not in line:
                continue
            key, content = line.strip().split(":")
            calib[key] = _rows_to_pose([float(v) for v in content.strip().split()])
    return calib


def parse_poses(filename: str, calibration: Dict[str, np.ndarray]) -> List[np.ndarray]:
    """poses.txt (camera frame) -> velodyne-frame poses Tr^-1 . P . Tr, float64 (semantickitti_ms.py:377-401)."""
    tr = calibration["Tr"]
    tr_inv = np.linalg.inv(tr)
    poses = []
    with open(filename) as f:
        for line in f:
            vals = [float(v) for v in line.strip().split()]
            if len(vals) >= 12:
                poses.append(np.matmul(tr_inv, np.matmul(_rows_to_pose(vals), tr)))
    return poses


class KittiSequence:
    """One sequence directory: frame paths + float32 velodyne-frame poses (the reference casts them, :343)."""

    def __init__(self, root: str, seq: int):
        self.dir = os.path.join(root, str(seq).zfill(2))
        calib = parse_calibration(os.path.join(self.dir, "calib.txt"))
        self.poses = [p.astype(np.float32) for p in parse_poses(os.path.join(self.dir, "poses.txt"), calib)]
        self.has_labels = os.path.isdir(os.path.join(self.dir, "labels"))

    def __len__(self):
        return len(self.poses)

    def scan_path(self, frame: int) -> str:
        return os.path.join(self.dir, "velodyne", str(frame).zfill(6) + ".bin")

    def points(self, frame: int) -> np.ndarray:
        return np.fromfile(self.scan_path(frame), dtype=np.float32).reshape((-1, 4))

    def raw_labels(self, frame: int, subdir: str = "labels") -> np.ndarray:
        """semantic ids (lower 16 bits) of a .label file; `subdir` may point at a prediction directory laid out like
        the dataset (the reference's pseudo labels, :295-299)."""
        path = os.path.join(self.dir, subdir, str(frame).zfill(6) + ".label")
        return (np.fromfile(path, dtype=np.uint32) & 0xFFFF).astype(np.int64)


def multiscan_sample(seq: KittiSequence, frame: int, multiscan: int, steps: Sequence[int], device="cuda",
                     pseudo_subdir: Optional[str] = None) -> Dict:
    """Frame `frame` and its up-to-`multiscan` history frames (ONLY_HISTORY, oldest first; frames before the start of
    the sequence are skipped like the reference's try/except, :285-291) as resident tensors for build_multiscan_batch.
    Pseudo labels: the annotations themselves (PSEUDO_MASK 'gt') or the .label files under `pseudo_subdir`."""
    dev = torch.device(device)
    frames = [frame + d for d in range(-multiscan, 0) if frame + d >= 0] + [frame]
    pts, labs, poses, pseudo = [], [], [], []
    for f in frames:
        raw = seq.raw_labels(f) if seq.has_labels else np.zeros(len(seq.points(f)), dtype=np.int64)
        pts.append(torch.from_numpy(seq.points(f)).to(dev))
        labs.append(torch.from_numpy(_LUT[raw]).to(dev))
        poses.append(torch.from_numpy(seq.poses[f]).to(dev))
        if f != frame:
            praw = raw if pseudo_subdir is None else seq.raw_labels(f, pseudo_subdir)
            pseudo.append(torch.from_numpy(_CANON[praw]).to(dev))
    return {"points": pts, "labels": labs, "poses": poses, "pseudo": pseudo, "deltas": [f - frame for f in frames[:-1]],
            "name": seq.scan_path(frame)}
