"""nuScenes multi-scan (FSA) data stage - the temporal aggregation of R/pcseg/data/dataset/nuscenes/nuscenes_ms.py:226-373
and the voxelisation of nuscenes_voxel_ms.py:77-187 on resident sweeps.

Split between host and device the way the work splits:

  host (numpy, a few dozen 3x3 products per sample, results cached per keyframe like the reference's
  `token2samplelist`):
      rotation_matrix      `Quaternion(q).rotation_matrix` of pyquaternion (nuscenes-devkit's quaternion class)
      relative_transform   (R, T) of transform_point (:348-373): lidar frame of `info` -> lidar frame of `info0`
      select_sweeps        which sweeps of the scene to aggregate (:238-276): nearest frame to every multiple of
                           STEP metres of driven distance up to MULTISCAN, plus every keyframe on the way
      sweep_params         per selected sweep the 28 doubles ts_fuse_sweeps consumes
  device (one launch per sample over all selected sweeps):
      ts_fuse_sweeps       ego-box filter on the raw coordinates, sensor -> keyframe -> current-frame transforms in
                           float64 with numpy's rounding points, time delta column
      class-step mask      table lookup on the pseudo labels: keep class c of the sweep at position p iff
                           steps[c] != 0 and (p + 1) % steps[c] == 0 (:322-327)
      voxelisation         the same ts_voxel_coords / ts_sparse_quantize stage as SemanticKITTI (voxel 0.1 m,
                           IN_FEATURE_DIM 4: the time delta column is cut, nuscenes fsa yaml:16,28)

A sequence is described by plain arrays (`NuscSequence`), not by devkit objects: the reference reads these numbers out
of its info pickles (mmdet3d layout) - file IO and the devkit are out of scope, the arithmetic on them is not.
Bit-exact against the reference's dataset code: tests/golden/multiscan_nus.npz.
"""
from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np
import torch

from .. import backend as B
from .stage import collate_batch, voxelize_sample_ms

__all__ = ["NuscSequence", "rotation_matrix", "relative_transform", "select_sweeps", "sweep_params", "fuse_sweeps",
           "build_nuscenes_batch"]


@dataclass
class NuscSequence:
    """Frames (keyframes and the sweeps between them) of one or more scenes in time order.

    per frame g:  is_key [F] bool, key_index [F] (row of the key_* arrays or -1), timestamps [F] int64 microseconds,
                  scene_tokens [F], local_indexes [F] (the keyframe whose lidar frame sensor2lidar_* maps into: the
                  next keyframe at or after g), s2l_r [F,3,3] / s2l_t [F,3] float64 (sweeps only)
    per keyframe: global_indexes [K] (its frame number), l2e_q / e2g_q [K,4] (w,x,y,z), l2e_t / e2g_t [K,3]"""
    is_key: np.ndarray
    key_index: np.ndarray
    timestamps: np.ndarray
    scene_tokens: Sequence
    local_indexes: np.ndarray
    s2l_r: np.ndarray
    s2l_t: np.ndarray
    global_indexes: np.ndarray
    l2e_q: np.ndarray
    l2e_t: np.ndarray
    e2g_q: np.ndarray
    e2g_t: np.ndarray


def rotation_matrix(q) -> np.ndarray:
    """pyquaternion's Quaternion(q).rotation_matrix (q = w, x, y, z; normalised unless unit to 1e-14): the lower-right
    3x3 of Q(q) Qbar(q)^T."""
    q = np.array(q, dtype=np.float64)
    n2 = float(q @ q)
    if abs(1.0 - n2) >= 1e-14 and n2 > 0:
        q = q / np.sqrt(n2)
    w, x, y, z = q
    left = np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])
    right = np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])
    return (left @ right.T)[1:, 1:]


def relative_transform(seq: NuscSequence, key0: int, key: int):
    """(R [3,3], T [3]) float64 with p_in_frame(key0) = p_in_frame(key) @ R + T (nuscenes_ms.py:348-373)."""
    l2e0, e2g0 = rotation_matrix(seq.l2e_q[key0]), rotation_matrix(seq.e2g_q[key0])
    l2e, e2g = rotation_matrix(seq.l2e_q[key]), rotation_matrix(seq.e2g_q[key])
    back = np.linalg.inv(e2g0).T @ np.linalg.inv(l2e0).T
    rot = (l2e.T @ e2g.T) @ back
    trans = (seq.l2e_t[key] @ e2g.T + seq.e2g_t[key]) @ back
    trans -= seq.e2g_t[key0] @ back + seq.l2e_t[key0] @ np.linalg.inv(l2e0).T
    return rot, trans


def select_sweeps(seq: NuscSequence, index: int, multiscan: int, step: float) -> List[int]:
    """Frame offsets (negative, ascending = oldest first) aggregated into keyframe `index` (nuscenes_ms.py:238-276)."""
    g0 = int(seq.global_indexes[index])
    offsets, dist, delta = [], [], 0
    while not dist or dist[-1] <= multiscan * step:
        delta -= 1
        g = g0 + delta                  # a negative g indexes from the end, as the reference's list does
        if seq.scene_tokens[g] != seq.scene_tokens[g0]:
            dist.append(1000)
            break
        origin = np.zeros((1, 3))
        if not seq.is_key[g]:
            origin = origin @ seq.s2l_r[g].T + seq.s2l_t[g]          # the sweep's sensor origin in its keyframe's frame
        father = int(seq.local_indexes[g])
        if father != index:
            rot, trans = relative_transform(seq, index, father)
            origin = origin @ rot + trans
        offsets.append(delta)
        dist.append(float(np.linalg.norm(origin.reshape(-1)[:2], ord=2)))
    picked, cur = [], 1
    for i in range(len(offsets)):
        if dist[i] - cur * step > 0 or (dist[i] < dist[i + 1] and abs(dist[i] - cur * step) < abs(dist[i + 1] - cur * step)):
            picked.append(offsets[i])
            cur += 1
        if cur > multiscan:
            break
    picked += [d for d in offsets if seq.is_key[g0 + d]]            # every keyframe on the way (:270-272)
    return sorted(set(picked))


def sweep_params(seq: NuscSequence, index: int, offsets: Sequence[int]) -> np.ndarray:
    """[S, 28] float64 for ts_fuse_sweeps: {A[9], a[3], flagA, B[9], b[3], flagB, dt, 0} per selected sweep."""
    g0 = int(seq.global_indexes[index])
    out = np.zeros((len(offsets), 28), dtype=np.float64)
    for i, d in enumerate(offsets):
        g = g0 + d
        if seq.is_key[g]:
            rot, trans = relative_transform(seq, index, int(seq.key_index[g]))
            out[i, 13:22], out[i, 22:25], out[i, 25] = rot.reshape(-1), trans, 1.0
        else:
            out[i, 0:9], out[i, 9:12], out[i, 12] = np.asarray(seq.s2l_r[g]).reshape(-1), seq.s2l_t[g], 1.0
            father = int(seq.local_indexes[g])
            if father != index:
                rot, trans = relative_transform(seq, index, father)
                out[i, 13:22], out[i, 22:25], out[i, 25] = rot.reshape(-1), trans, 1.0
        out[i, 26] = seq.timestamps[g0] / 1e6 - seq.timestamps[g] / 1e6
    return out


_tables = {}


def _layout(lengths, steps, device):
    key = (tuple(lengths), tuple(steps), str(device))
    hit = _tables.get(key)
    if hit is None:
        if len(_tables) >= 64:
            _tables.pop(next(iter(_tables)))
        idx = torch.repeat_interleave(torch.arange(len(lengths), dtype=torch.int32), torch.tensor(list(lengths))).to(device)
        table = torch.tensor([[bool(st) and (pos + 1) % st == 0 for st in steps] for pos in range(len(lengths))],
                             dtype=torch.bool, device=device)
        hit = (idx, table)
        _tables[key] = hit
    return hit


def fuse_sweeps(cur_pts, cur_lab, hist_pts: List[torch.Tensor], hist_lab: List[torch.Tensor],
                hist_pseudo: List[torch.Tensor], params: torch.Tensor, steps: Sequence[int]):
    """Current keyframe + selected sweeps -> the un-filtered stack (raw [n, 5] = x, y, z, intensity, time delta; labels
    [n]; keep [n]).  `raw[keep]` is the reference's `xyzret_ms` (nuscenes_ms.py:125-127): current scan first (time
    column 0, :109), then the sweeps oldest first, each without its ego-box points and filtered by the class-step rule.
    hist_lab[i]: mapped labels of a keyframe, zeros for a sweep (:297, :318); hist_pseudo[i]: pseudo labels (:323)."""
    dev = cur_pts.device
    cur = cur_pts.clone()
    cur[:, 4] = 0
    n_cur = cur.shape[0]
    if not hist_pts:
        return cur, cur_lab.long(), torch.ones(n_cur, dtype=torch.bool, device=dev)
    sweep_idx, table = _layout([p.shape[0] for p in hist_pts], steps, dev)
    stack = torch.cat(hist_pts, 0).contiguous()
    pseudo = torch.cat(hist_pseudo, 0).long()
    fused, no_ego = B.fuse_sweeps(stack, sweep_idx, params)
    keep = no_ego & table.view(-1)[sweep_idx.long() * table.shape[1] + pseudo]
    raw = torch.cat([cur, fused], 0)
    lab = torch.cat([cur_lab.long(), torch.cat(hist_lab, 0).long()])
    return raw, lab, torch.cat([torch.ones(n_cur, dtype=torch.bool, device=dev), keep])


def build_nuscenes_batch(samples: List[Dict], voxel_size: float, steps: Sequence[int], in_feature_dim: int = 4) -> Dict:
    """samples[b] = dict(points [n,5], labels [n], hist_points [..], hist_labels [..], hist_pseudo [..],
    params [S,28] float64 tensor, name).  Returns the collated batch_dict MinkUNetMs consumes
    (nuscenes_voxel_ms.py:77-212 == the SemanticKITTI stage on the first `in_feature_dim` columns).
    The whole batch goes through ONE chain of launches: one ts_fuse_sweeps over every sweep point of every sample (ego box,
    sensor -> keyframe -> current-frame transforms, time delta), the class-step rule as one table lookup, then
    stage.voxelize_batch_ms (one compaction, one batch-keyed voxelisation per cloud kind)."""
    from . import stage as _stage
    if not _stage._BATCHED or not samples or len(samples) > 64:
        return build_nuscenes_batch_per_sample(samples, voxel_size, steps, in_feature_dim)
    dev = samples[0]["points"].device
    f = in_feature_dim
    n_cls = len(steps)
    cur_all = torch.cat([s["points"] for s in samples], 0)          # (a fresh tensor: the resident scans stay untouched)
    cur_all[:, 4] = 0                                                 # time column of the current keyframe (:109)
    n_cur = [int(s["points"].shape[0]) for s in samples]
    cur_f = cur_all[:, :f].contiguous()
    cuts = [0]
    for n in n_cur:
        cuts.append(cuts[-1] + n)
    cur_list = [cur_f[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    lab_list = [s["labels"].long() for s in samples]
    hp, hl, hs, lengths, sample_of_sweep, rows, params = [], [], [], [], [], [], []
    for b, s in enumerate(samples):
        for pos, p in enumerate(s["hist_points"]):
            hp.append(p)
            lengths.append(int(p.shape[0]))
            sample_of_sweep.append(b)
            rows.append([bool(st) and (pos + 1) % st == 0 for st in steps])           # nuscenes_ms.py:320-328
        hl += list(s["hist_labels"])
        hs += list(s["hist_pseudo"])
        if len(s["hist_points"]):
            params.append(s["params"])
    if hp:
        stack = torch.cat(hp, 0).contiguous()
        pseudo = torch.cat(hs, 0).long()
        lab_h = torch.cat(hl, 0).long()
        sweep32 = _stage.rows_index32(lengths, dev)
        key = ("nusc-table", tuple(map(tuple, rows)), tuple(sample_of_sweep), str(dev))
        hit = _tables.get(key)
        if hit is None:
            if len(_tables) >= 64:
                _tables.pop(next(iter(_tables)))
            hit = (torch.tensor(rows, dtype=torch.bool).to(dev), torch.tensor(sample_of_sweep, dtype=torch.int64).to(dev))
            _tables[key] = hit
        table, sample_of = hit
        fused, no_ego = B.fuse_sweeps(stack, sweep32, torch.cat(params, 0) if len(params) > 1 else params[0])
        hist_ms = fused[:, :f].contiguous()
    else:
        hist_ms = torch.empty((0, f), dtype=cur_f.dtype, device=dev)
        lab_h = pseudo = torch.empty(0, dtype=torch.int64, device=dev)
        sweep32 = torch.empty(0, dtype=torch.int32, device=dev)
        no_ego = None
        table = torch.zeros((1, n_cls), dtype=torch.bool, device=dev)
        sample_of = torch.zeros(1, dtype=torch.int64, device=dev)
    return _stage.voxelize_batch_ms(cur_list, lab_list, cur_f, hist_ms, lab_h, sweep32, pseudo, table, sample_of, voxel_size,
                                    [s.get("name", "") for s in samples], pre_keep=no_ego)


def build_nuscenes_batch_per_sample(samples: List[Dict], voxel_size: float, steps: Sequence[int], in_feature_dim: int = 4) -> Dict:
    """build_nuscenes_batch sample by sample (the form the batched stage replaced; its cross-check and TASEG_STAGE_BATCHED=0)"""
    out = []
    for s in samples:
        raw, lab, keep = fuse_sweeps(s["points"], s["labels"], s["hist_points"], s["hist_labels"], s["hist_pseudo"],
                                     s["params"], steps)
        cur = s["points"].clone()
        cur[:, 4] = 0
        out.append(voxelize_sample_ms(cur[:, :in_feature_dim].contiguous(), s["labels"].long(),
                                      raw[:, :in_feature_dim].contiguous(), lab, voxel_size, s.get("name", ""), keep=keep))
    return collate_batch(out)
