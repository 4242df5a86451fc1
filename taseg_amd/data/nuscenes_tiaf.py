"""nuScenes TIAF (temporal image aggregation and fusion) data stage on the device - the camera side of
R/pcseg/data/dataset/nuscenes/nuscenes_ms_mm.py:196-401 and the voxelisation / collate of
nuscenes_voxel_ms_mm.py:77-262, next to the nuScenes FSA stage of taseg_amd.data.nuscenes:

  host (a few 3x3 products per image keyframe)
      select_image_keyframes   which keyframes' camera images join the stack (:204-236): the keyframe nearest to every
                               STEP_IMAGE metres of driven distance up to MULTISCAN_IMAGE, filled up from the passed-over
                               ones with random.sample (the caller's `rng`), plus the current one
      camera_chain             the 57 doubles ts_project_cam consumes: lidar -> ego -> global -> camera ego -> camera
                               (calibrated-sensor / ego-pose records, :349-367) and the intrinsic
  device, per image keyframe
      ts_fuse_sweeps           the keyframe's points and those of its MULTISCAN_INTERVAL predecessors in ITS lidar frame,
                               ego box cut, time delta to the current keyframe (:246-295); paint radius (:297-300)
      ts_project_cam           per used camera: projection, in-image test, half-resolution pixel, top rows cut, the row
                               shifted by HEIGHT * (position of the image in the stack)            (:349-398)
      ts_fuse_sweeps           the kept points into the current lidar frame (:305; float64 product rounded once)
      image                    half-resolution uint8 RGB -> float32 BGR / 255, the two top rows cut (:381-386)
  sample / batch               the three clouds (current, fused, FOV) voxelised with ONE coordinate shift, the FOV cloud
                               clamped to the current cloud's corner; sparse collate + image stacks as NCHW + offset_img

File decoding (JPEG, .npy semantic maps) and PIL's bilinear half-size resize (:380) stay on the host, as in the reference:
the stage starts from resident half-resolution uint8 images.  Bit-exact against the reference's dataset code:
tests/golden/tiaf_nus.npz.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import backend as B
from ..torchsparse import SparseTensor
from .nuscenes import NuscSequence, fuse_sweeps, relative_transform, rotation_matrix
from .stage import _quantize, collate_batch, voxelize_sample_ms

__all__ = ["select_image_keyframes", "camera_chain", "frame_cloud", "fov_points", "half_image", "build_nusc_tiaf_sample",
           "build_nusc_tiaf_batch"]


def _scene_of(seq: NuscSequence, key: int):
    return seq.scene_tokens[int(seq.global_indexes[key])]


def select_image_keyframes(seq: NuscSequence, index: int, multiscan_image: int, step_image: float, rng) -> List[int]:
    """Keyframe offsets (ascending, the current one = 0 last) whose images are stacked (nuscenes_ms_mm.py:204-236).
    `rng.sample` plays the part of the reference's global random.sample."""
    if multiscan_image == 0:
        return [0]
    n_keys = len(seq.global_indexes)
    delta, total, dist = 0, [], []
    while not dist or dist[-1] <= multiscan_image * step_image:
        delta -= 1
        other = (index + delta) % n_keys            # a negative index wraps, as the reference's list indexing does
        if _scene_of(seq, other) != _scene_of(seq, index):
            dist.append(1000)
            break
        _, trans = relative_transform(seq, index, other)       # the lidar origin of that keyframe in the current frame
        total.append(delta)
        dist.append(float(np.linalg.norm(trans[:2], ord=2)))
    cur, picked, passed = 1, [], []
    for i in range(len(total)):
        if dist[i] - cur * step_image > 0 or (dist[i] < dist[i + 1] and abs(dist[i] - cur * step_image) < abs(dist[i + 1] - cur * step_image)):
            picked.append(total[i])
            cur += 1
        else:
            passed.append(total[i])
        if cur > multiscan_image:
            break
    if len(picked) < multiscan_image and passed:
        picked += rng.sample(passed, min(multiscan_image - len(picked), len(passed)))
    return sorted(set(picked)) + [0]


def camera_chain(lidar_cs_q, lidar_cs_t, lidar_pose_q, lidar_pose_t, cam_pose_q, cam_pose_t, cam_cs_q, cam_cs_t,
                 intrinsic) -> np.ndarray:
    """[57] float64 for ts_project_cam (layout: include/taseg_hip.h)."""
    out = np.zeros(57, dtype=np.float64)
    out[0:9] = rotation_matrix(lidar_cs_q).reshape(-1)
    out[9:12] = lidar_cs_t
    out[12:21] = rotation_matrix(lidar_pose_q).reshape(-1)
    out[21:24] = lidar_pose_t
    out[24:27] = cam_pose_t
    out[27:36] = np.ascontiguousarray(rotation_matrix(cam_pose_q).T).reshape(-1)
    out[36:39] = cam_cs_t
    out[39:48] = np.ascontiguousarray(rotation_matrix(cam_cs_q).T).reshape(-1)
    out[48:57] = np.asarray(intrinsic, dtype=np.float64).reshape(-1)
    return out


def _move(points5: torch.Tensor, seq: NuscSequence, key0: int, frames: Sequence[int], lengths: Sequence[int], stamp0: int):
    """ts_fuse_sweeps over the concatenated clouds of keyframes `frames` (rows of `lengths`): frame f into the lidar frame of
    key0 (no product for f == key0... unless `always`), column 4 = time delta to `stamp0`; returns (moved, outside-ego-box)"""
    params = np.zeros((len(frames), 28), dtype=np.float64)
    for i, f in enumerate(frames):
        if f != key0:
            rot, trans = relative_transform(seq, key0, f)
            params[i, 13:22], params[i, 22:25], params[i, 25] = rot.reshape(-1), trans, 1.0
        params[i, 26] = stamp0 / 1e6 - seq.timestamps[int(seq.global_indexes[f])] / 1e6
    dev = points5.device
    idx = torch.repeat_interleave(torch.arange(len(frames), dtype=torch.int32), torch.tensor(list(lengths))).to(dev)
    return B.fuse_sweeps(points5.contiguous(), idx, torch.from_numpy(params).to(dev))


def frame_cloud(seq: NuscSequence, index: int, delta: int, prev_delta: Optional[int], interval: int,
                key_points: Dict[int, torch.Tensor], key_labels: Dict[int, torch.Tensor], paint_dist: float):
    """The cloud image keyframe index + delta projects (nuscenes_ms_mm.py:246-300): (raw [m,5] float32 in ITS lidar frame,
    labels [m])."""
    i = index + delta
    frames = [i]
    for meta in range(-interval, 0):
        j = i + meta
        if j < 0 or j >= len(seq.global_indexes) or _scene_of(seq, j) != _scene_of(seq, index):
            continue
        if prev_delta is not None and delta + meta <= prev_delta:
            continue
        frames.append(j)
    pts = torch.cat([key_points[f] for f in frames], 0)
    lab = torch.cat([key_labels[f].long() for f in frames], 0)
    moved, no_ego = _move(pts, seq, i, frames, [key_points[f].shape[0] for f in frames],
                          int(seq.timestamps[int(seq.global_indexes[index])]))
    moved, lab = moved[no_ego], lab[no_ego]
    if paint_dist > 0:
        radius = torch.sqrt(moved[:, 0] * moved[:, 0] + moved[:, 1] * moved[:, 1])
        near = radius <= paint_dist
        moved, lab = moved[near], lab[near]
    return moved, lab


def fov_points(cloud: torch.Tensor, labels: torch.Tensor, cam: torch.Tensor, image_size, crop_top: int, height: int,
               img_batch: int):
    """get_fov_points (:329-401) on the device: ([m, 6] = x, y, z, intensity, row + height * img_batch, col; labels [m])."""
    pts = cloud[:, :4].contiguous()
    pix, keep = B.project_cam(pts, cam, image_size, crop_top, float(height * img_batch))
    return torch.cat([pts[keep], pix[keep]], 1), labels[keep]


_lut = {}


def half_image(image_u8: torch.Tensor, crop_top: int = 2) -> torch.Tensor:
    """[h, w, 3] uint8 RGB at half resolution -> float32 BGR / 255 without the top rows (:381-386); the 256 quotients come
    from a table of correctly rounded float32 divisions (see taseg_amd.data.tiaf.crop_image)."""
    if image_u8.dtype != torch.uint8:
        raise TypeError("camera images must be uint8")
    lut = _lut.get(image_u8.device)
    if lut is None:
        lut = _lut[image_u8.device] = torch.from_numpy(np.arange(256, dtype=np.float32) / 255.).to(image_u8.device)
    return lut[image_u8.flip(2).long()][crop_top:].contiguous()


def build_nusc_tiaf_sample(fsa: Dict, seq: NuscSequence, index: int, key_points: Dict[int, torch.Tensor],
                           key_labels: Dict[int, torch.Tensor], lidar_cs, view_cs: Dict[int, tuple],
                           cam_frames: Dict[tuple, Dict], steps: Sequence[int], multiscan_image: int, step_image: float,
                           interval: int, used_view: Sequence[int], paint_dist: float, full_image_size, voxel_size: float,
                           rng, in_feature_dim: int = 4, crop_top: int = 2, name: str = "") -> Dict:
    """One sample of NuscVoxelMsMmDataset (nuscenes_voxel_ms_mm.py:77-223) from resident tensors.

    fsa          the FSA inputs of the keyframe, as taseg_amd.data.nuscenes.build_nuscenes_batch takes them (points, labels,
                 hist_points / hist_labels / hist_pseudo, params)
    key_points / key_labels   {keyframe: raw [n,5] float32 / mapped labels} of the keyframes the image side touches
    lidar_cs     (rotation q, translation) of the lidar's calibrated sensor; the lidar's ego pose = the keyframe's e2g pose
    view_cs      {view: (rotation q, translation, intrinsic [3,3])} of the cameras' calibrated sensors
    cam_frames   {(keyframe, view): dict(pose_q, pose_t, image uint8 [h,w,3] RGB at HALF resolution, semantic [h,w,1])}
    full_image_size   (W, H) of the camera images before the half-size resize"""
    frames = select_image_keyframes(seq, index, multiscan_image, step_image, rng)
    w_full, h_full = full_image_size
    height = h_full // 2 - crop_top
    dev = fsa["points"].device
    fov, fov_lab, images, semantic = [], [], [], []
    for batch_idx, d in enumerate(frames):
        i = index + d
        if i < 0 or i >= len(seq.global_indexes) or _scene_of(seq, i) != _scene_of(seq, index):
            continue
        cloud, lab = frame_cloud(seq, index, d, frames[batch_idx - 1] if batch_idx > 0 else None, interval, key_points,
                                 key_labels, paint_dist)
        for view_idx, v in enumerate(used_view):
            rec = cam_frames[(i, v)]
            chain = camera_chain(lidar_cs[0], lidar_cs[1], seq.e2g_q[i], seq.e2g_t[i], rec["pose_q"], rec["pose_t"],
                                 view_cs[v][0], view_cs[v][1], view_cs[v][2])
            pts, pl = fov_points(cloud, lab, torch.from_numpy(chain).to(dev), (w_full, h_full), crop_top, height,
                                 batch_idx * len(used_view) + view_idx)
            # into the current lidar frame - the current keyframe too: its (R, T) is a rounded identity (:305)
            rot, trans = relative_transform(seq, index, i)
            params = np.zeros((1, 28), dtype=np.float64)
            params[0, 13:22], params[0, 22:25], params[0, 25] = rot.reshape(-1), trans, 1.0
            five = torch.cat([pts[:, :4], torch.zeros((pts.shape[0], 1), dtype=torch.float32, device=dev)], 1).contiguous()
            moved, _ = B.fuse_sweeps(five, torch.zeros(pts.shape[0], dtype=torch.int32, device=dev), torch.from_numpy(params).to(dev))
            fov.append(torch.cat([moved[:, :4], pts[:, 4:]], 1))
            fov_lab.append(pl)
            images.append(half_image(rec["image"], crop_top))
            semantic.append(rec["semantic"].float()[crop_top:].contiguous())
    fov, fov_lab = torch.cat(fov, 0), torch.cat(fov_lab, 0)
    fov_all, fov_lab_all = fov, fov_lab                                  # the reference's xyzret_fov_ms / labels_fov_ms
    # the single-frame and fused clouds: the FSA stage (voxel_ms_mm.py:80-87, 127-186 == nuscenes_voxel_ms.py)
    raw, lab_all, keep = fuse_sweeps(fsa["points"], fsa["labels"], fsa["hist_points"], fsa["hist_labels"], fsa["hist_pseudo"],
                                     fsa["params"], steps)
    cur = fsa["points"].clone()
    cur[:, 4] = 0
    point = cur[:, :in_feature_dim].contiguous()
    sample = voxelize_sample_ms(point, fsa["labels"].long(), raw[:, :in_feature_dim].contiguous(), lab_all, voxel_size, name,
                                keep=keep, return_shift=True)
    lo = point[:, :3].t().contiguous().min(1).values
    inside = (fov[:, :3] >= lo).all(1)                                   # clamp_fov_mask (:133-135)
    fov, fov_lab = fov[inside].contiguous(), fov_lab[inside]
    shift = sample.pop("_shift")
    pc_fov, _, inds_fov, _ = _quantize(fov, voxel_size, shift=shift)
    sample["lidar_fov_ms"] = SparseTensor(fov[inds_fov], pc_fov[inds_fov])
    # (the reference pairs the FOV labels with the FUSED cloud's coordinates, :204 - kept as it is)
    sample["targets_fov_ms"] = SparseTensor(fov_lab[inds_fov], sample["lidar_ms"].C)
    sample["image_ms"] = torch.stack(images, 0)
    sample["semantic_map_ms"] = torch.stack(semantic, 0)
    n_img = len(images)
    sample["depth_map_ms"] = torch.zeros((n_img, h_full - crop_top, w_full, 1), dtype=torch.float32, device=dev)   # (:346-347)
    sample["lidar_map_ms"] = torch.zeros((n_img, h_full - crop_top, w_full, 4), dtype=torch.float32, device=dev)
    sample["_image_keyframes"], sample["_fov_points"], sample["_fov_labels"] = frames, fov_all, fov_lab_all   # (diagnostics)
    return sample


def build_nusc_tiaf_batch(samples: List[Dict]) -> Dict:
    """collate_batch of nuscenes_voxel_ms_mm.py:225-262 on device tensors."""
    stacks = {k: [s.pop(k) for s in samples] for k in ("image_ms", "depth_map_ms", "lidar_map_ms", "semantic_map_ms")}
    for s in samples:
        for k in ("_image_keyframes", "_fov_points", "_fov_labels"):
            s.pop(k, None)
    out = collate_batch(samples)
    dev = stacks["image_ms"][0].device
    out["offset_img"] = torch.cumsum(torch.tensor([i.shape[0] for i in stacks["image_ms"]]), 0).int().to(dev)
    for k, v in stacks.items():
        out[k] = torch.cat(v, 0).permute(0, 3, 1, 2).contiguous()
    return out
