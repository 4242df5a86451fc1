"""TIAF (temporal image aggregation and fusion) data stage on the device - the camera side of
R/pcseg/data/dataset/semantickitti/semantickitti_ms_mm.py:304-461 and the voxelisation / collate of
semantickitti_voxel_ms_mm.py:79-266, next to the LiDAR multi-scan stage of taseg_amd.data.stage:

  per frame with an image (the current one and every STEP_IMAGE-th history frame up to MULTISCAN_IMAGE):
      ts_project_fov    points in front of the camera -> pixel (row, col) through P2 @ Tr, frustum + crop test, the row
                        shifted by HEIGHT * (position of the frame in the image stack)           (:419-457)
      ts_fuse_scans     the kept points into the current frame (history frames only)              (:377)
      image             uint8 RGB -> float32 BGR / 255, top-left crop, zero padded to HEIGHT x WIDTH (:432-447)
  sample                ring id column of the single-frame cloud (:131-141), the three clouds (current, fused, FOV)
                        voxelised with ONE coordinate shift (the fused cloud's minimum), FOV cloud clamped to the current
                        cloud's corner like the fused one                                       (voxel_ms_mm.py:124-204)
  batch                 sparse collate + image stacks concatenated along the frame axis as NCHW + offset_img (:223-266)

File decoding (PNG, .npy semantic maps) and the optional colour jitter / flips are outside the scope contract; the stage
starts from resident uint8 images.  Bit-exact against the reference's dataset code: tests/golden/tiaf_data.npz.
"""
from typing import Dict, List, Sequence

import torch

from .. import backend as B
from ..torchsparse import SparseTensor
from .stage import _fuse_history, _quantize, collate_batch, voxelize_sample_ms

__all__ = ["ring_id", "fov_points", "crop_image", "build_tiaf_sample", "build_tiaf_batch"]


def ring_id(points: torch.Tensor) -> torch.Tensor:
    """get_kitti_points_ringID (semantickitti_ms_mm.py:131-141) on the device: float32 [n]."""
    yaw = -torch.atan2(points[:, 1], -points[:, 0])
    proj_x = 0.5 * (yaw / torch.pi + 1.0)
    wrap = torch.zeros_like(proj_x)
    wrap[1:] = ((proj_x[1:] < 0.2) & (proj_x[:-1] > 0.8)).to(proj_x.dtype)
    return torch.clamp(torch.cumsum(wrap, 0), 0, 63)


def fov_points(points: torch.Tensor, proj: torch.Tensor, image_size, crop, img_batch: int) -> torch.Tensor:
    """[m, 6] = (x, y, z, intensity, row + HEIGHT * img_batch, col) of the points that project into the cropped image."""
    pts = points[:, :4].contiguous()
    pix, keep = B.project_fov(pts, proj, image_size, crop, float(crop[0] * img_batch))
    return torch.cat([pts[keep], pix[keep]], 1)


_lut = {}


def crop_image(image_u8: torch.Tensor, crop) -> torch.Tensor:
    """[h, w, 3] uint8 RGB -> float32 [HEIGHT, WIDTH, 3] BGR / 255, zero padded (semantickitti_ms_mm.py:432-447).  The 256
    possible quotients come from a table of correctly rounded float32 divisions (numpy's `image / 255.`): the device's
    division by a scalar is a multiplication by the reciprocal and differs in the last bit."""
    if image_u8.dtype != torch.uint8:
        raise TypeError("camera images must be uint8")
    lut = _lut.get(image_u8.device)
    if lut is None:
        import numpy as np
        lut = _lut[image_u8.device] = torch.from_numpy(np.arange(256, dtype=np.float32) / 255.).to(image_u8.device)
    return _pad(lut[image_u8.flip(2).long()], crop)


def build_tiaf_sample(frames: Dict[int, Dict], steps: Sequence[int], multiscan: int, step_image: int, proj: torch.Tensor,
                      crop, voxel_size: float, name: str = "", fov_dist: float = -1.0) -> Dict:
    """frames[delta] (delta = 0 current, < 0 history) = dict(points [n,4], labels [n] classes, pseudo [n] canonical classes
    or -1, pose [4,4] float32, and - for the frames with |delta| % step_image == 0 - image [h,w,3] uint8 RGB,
    semantic [h,w,1]).  Returns the reference's sample dictionary (semantickitti_voxel_ms_mm.py:205-227)."""
    cur = frames[0]
    pose0 = cur["pose"]
    hist = [d for d in sorted(frames) if -multiscan <= d < 0]
    raw_all, lab_all, keep = _fuse_history(cur["points"], cur["labels"], [frames[d]["points"] for d in hist],
                                           [frames[d]["labels"] for d in hist], pose0, [frames[d]["pose"] for d in hist], hist,
                                           steps, [frames[d]["pseudo"] for d in hist])
    # single-frame features carry the ring id as 5th column (:266-267), the fused cloud the time flag (:159)
    point = torch.cat([cur["points"][:, :4], ring_id(cur["points"]).unsqueeze(1)], 1).contiguous()
    sample = voxelize_sample_ms(point, cur["labels"].long(), raw_all, lab_all, voxel_size, name, keep=keep, return_shift=True)
    # camera frames, newest first (the reference inserts at the front while walking delta upwards, :378-382)
    fov, images, semantic = [], [], []
    for d in sorted((d for d in frames if "image" in frames[d]), reverse=True):
        f = frames[d]
        h, w = f["image"].shape[0], f["image"].shape[1]
        pts = fov_points(f["points"], proj, (w, h), crop, abs(d) // step_image)
        if fov_dist > 0:
            radius = torch.sqrt(pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1])
            pts = pts[radius <= fov_dist]
        if d != 0:
            moved = B.fuse_scan(pts[:, :4].contiguous(), pose0, f["pose"])
            pts = torch.cat([moved, pts[:, 4:]], 1)
        fov.append(pts)
        images.append(crop_image(f["image"], crop))
        semantic.append(_pad(f["semantic"].float(), crop))
    fov = torch.cat(fov, 0)
    # FOV cloud: clamped to the current cloud's corner, rounded and shifted like the other two, grouped per voxel
    lo = point[:, :3].t().contiguous().min(1).values       # (row-wise minimum of the transposed copy: see data/stage.py)
    fov = fov[(fov[:, :3] >= lo).all(1)].contiguous()
    shift = sample.pop("_shift")
    pc_fov, _, inds_fov, _ = _quantize(fov, voxel_size, shift=shift)
    sample["lidar_fov_ms"] = SparseTensor(fov[inds_fov], pc_fov[inds_fov])
    sample["image_ms"] = torch.stack(images, 0)
    sample["semantic_map_ms"] = torch.stack(semantic, 0)
    return sample


def _pad(t: torch.Tensor, crop) -> torch.Tensor:
    out = torch.zeros((crop[0], crop[1], t.shape[2]), dtype=torch.float32, device=t.device)
    r, c = min(crop[0], t.shape[0]), min(crop[1], t.shape[1])
    out[:r, :c] = t[:r, :c]
    return out


def build_tiaf_batch(samples: List[Dict]) -> Dict:
    """collate_batch of semantickitti_voxel_ms_mm.py:223-266 on device tensors."""
    images = [s.pop("image_ms") for s in samples]
    semantic = [s.pop("semantic_map_ms") for s in samples]
    out = collate_batch(samples)
    dev = images[0].device
    out["offset_img"] = torch.cumsum(torch.tensor([i.shape[0] for i in images]), 0).int().to(dev)
    out["image_ms"] = torch.cat(images, 0).permute(0, 3, 1, 2).contiguous()
    out["semantic_map_ms"] = torch.cat(semantic, 0).permute(0, 3, 1, 2).contiguous()
    return out
