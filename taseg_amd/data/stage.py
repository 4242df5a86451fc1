"""Device-resident data stage: TASeg's multi-scan temporal aggregation (FSA) + dataset voxelisation.

The reference does this in numpy inside DataLoader workers
(R/pcseg/data/dataset/semantickitti/semantickitti_ms.py:140-149,253-320,403-417 and
semantickitti_voxel_ms.py:77-212); here the same steps run on the GPU on resident scans:

  pose fuse            ts_fuse_scan        p' = ((p R_t^T + t_t) - t_0) R_0, float32, reference summation order
  class-step filter    lookup table        keep a history point iff steps[class] != 0 and |delta| % steps[class] == 0
  concat + time flag   torch.cat           current scan first (flag 1), kept history points after it (flag 0)
  voxel coordinates    ts_voxel_coords     int32(round_half_even(xyz / voxel)) - min
  voxel grouping       ts_sparse_quantize  radix sort + first-occurrence representative + inverse map

Outputs have the reference's exact layout and ORDER (batch_dict schema of SURVEY.md appendix C), checked bit
for bit against the reference dataset code in tests (tests/golden/multiscan.npz).
"""
from typing import Dict, List, Sequence

import torch

from .. import backend as B
from ..torchsparse import SparseTensor
from ..options import options

__all__ = ["fuse_multiscan", "voxelize_sample_ms", "voxelize_sample", "collate_batch", "build_multiscan_batch",
           "build_multiscan_batch_per_sample", "voxelize_batch_ms", "rows_index", "DevicePrefetcher"]


_static_cache = {}


def _history_index(lengths: Sequence[int], deltas: Sequence[int], steps: Sequence[int], device):
    """Per-layout helpers that depend only on the scan lengths and the schedule, cached: scan index of every
    concatenated history point, and the [T, C] table "is class c aggregated from the scan delta frames away"
    (semantickitti_ms.py:303-308: steps[c] != 0 and |delta| % steps[c] == 0)."""
    key = (tuple(lengths), tuple(deltas), tuple(steps), str(device))
    hit = _static_cache.get(key)
    if hit is None:
        if len(_static_cache) >= 64:
            _static_cache.pop(next(iter(_static_cache)))
        scan_idx = torch.repeat_interleave(torch.arange(len(lengths), dtype=torch.int32),
                                           torch.tensor(list(lengths), dtype=torch.int64)).to(device)
        # last column: pseudo class -1 = a label that is no class's canonical raw id (e.g. a moving-object id): never kept
        table = torch.tensor([[bool(st) and abs(d) % st == 0 for st in steps] + [False] for d in deltas], dtype=torch.bool)
        hit = (scan_idx, table.to(device))
        _static_cache[key] = hit
    return hit


def _fuse_history(cur_pts, cur_lab, hist_pts, hist_lab, pose0, hist_poses, deltas, steps, hist_pseudo=None):
    """All history scans in one pass: concatenate, transform every point with its scan's pose (ts_fuse_scans),
    look the class-step decision up per point.  Returns the un-filtered stack [current | history] (x, y, z,
    intensity, time flag), its labels and the keep mask; order = current scan first, then history oldest first, each
    in file order - the order the reference's loop produces (semantickitti_ms.py:140-149).

    hist_pseudo[i] (optional): the class whose CANONICAL raw id the point's pseudo label is, or -1 - the reference
    compares the raw pseudo label with LEARNING_MAP_INV[class] (semantickitti_ms.py:303-308), so a raw id that maps to
    a class without being its canonical id (the moving-object ids 252 .. 259) is never aggregated.  Default: the
    annotation classes `hist_lab` (exact when labels only hold canonical ids, as the synthetic scans do)."""
    dev = cur_pts.device
    n_cur = cur_pts.shape[0]
    if len(hist_pts) == 0:
        flag = torch.ones((n_cur, 1), dtype=cur_pts.dtype, device=dev)
        return torch.cat([cur_pts[:, :4], flag], 1), cur_lab.long(), torch.ones(n_cur, dtype=torch.bool, device=dev)
    scan_idx, table = _history_index([p.shape[0] for p in hist_pts], deltas, steps, dev)
    hp = torch.cat([p[:, :4] for p in hist_pts], 0).contiguous()
    hl = torch.cat(hist_lab, 0).long()
    fused = B.fuse_scans(hp, scan_idx, pose0, torch.stack(list(hist_poses), 0))
    ps = hl if hist_pseudo is None else torch.cat(hist_pseudo, 0).long()
    ps = torch.where(ps < 0, torch.full_like(ps, table.shape[1] - 1), ps)
    keep = table.view(-1)[scan_idx.long() * table.shape[1] + ps]
    pts = torch.cat([cur_pts[:, :4], fused], 0)
    flag = torch.zeros((pts.shape[0], 1), dtype=pts.dtype, device=dev)
    flag[:n_cur] = 1                               # append_time_flag (semantickitti_ms.py:253-257)
    mask = torch.cat([torch.ones(n_cur, dtype=torch.bool, device=dev), keep])
    return torch.cat([pts, flag], 1), torch.cat([cur_lab.long(), hl]), mask


def fuse_multiscan(cur_pts, cur_lab, hist_pts: List[torch.Tensor], hist_lab: List[torch.Tensor], pose0,
                   hist_poses: List[torch.Tensor], deltas: Sequence[int], steps: Sequence[int]):
    """Current scan + filtered, pose-aligned history scans -> (raw_data_ms [n, 5], labels_ms [n]).

    cur_pts / hist_pts[i]: float32 [n, 4] (x, y, z, intensity) in their own sensor frames;
    *_lab: class ids (learning-map ids, 0..C-1); poses: 4x4 float32 (world <- sensor);
    deltas[i] < 0 is the frame offset of hist_pts[i].  History order is preserved (oldest first, as the
    reference iterates delta = -MULTISCAN .. -1)."""
    raw, lab, mask = _fuse_history(cur_pts, cur_lab, hist_pts, hist_lab, pose0, hist_poses, deltas, steps)
    return raw[mask], lab[mask]


def _quantize(points, voxel_size, shift=None):
    coords4, mins = B.voxel_coords(points, voxel_size, shift=shift)
    index, inverse = B.sparse_quantize(coords4)
    return coords4[:, :3], mins, index.long(), inverse.long()


def voxelize_sample(points, labels, voxel_size, name="") -> Dict:
    """Single-frame sample (semantickitti_voxel.py:119-150)."""
    pc, _, inds, inverse = _quantize(points, voxel_size)
    return {"name": name, "lidar": SparseTensor(points[inds], pc[inds]), "targets": SparseTensor(labels[inds], pc[inds]),
            "targets_mapped": SparseTensor(labels, pc), "inverse_map": SparseTensor(inverse, pc),
            "num_points": torch.tensor([points.shape[0]])}


def voxelize_sample_ms(points, labels, points_ms, labels_ms, voxel_size, name="", keep=None, return_shift=False) -> Dict:
    """Multi-scan sample (semantickitti_voxel_ms.py:121-187): both clouds voxelised, the single-frame one
    shifted by the fused cloud's minimum.  `keep` (optional bool mask over points_ms) is AND-ed with the clamp
    so the class-step filter and the clamp cost one compaction (one host read) instead of two."""
    # (min over dim 0 of the [n, 3] slice runs in one of torch's slow few-column reductions: 175 us for 35k points;
    #  the same minimum along the rows of the transposed copy takes ~10 us)
    lo = points[:, :3].t().contiguous().min(1).values
    clamp = (points_ms[:, :3] >= lo).all(1)                               # :121-124
    if keep is not None:
        clamp = clamp & keep
    points_ms, labels_ms = points_ms[clamp].contiguous(), labels_ms[clamp]
    pc_ms, mins_ms, inds_ms, inverse_ms = _quantize(points_ms, voxel_size)
    pc, _, inds, inverse = _quantize(points, voxel_size, shift=mins_ms)   # pc_ -= pc_ms_.min(0)  (:130)
    extra = {"_shift": mins_ms} if return_shift else {}      # the fused cloud's minimum: further clouds of the sample share it
    return {
        **extra,
        "name": name,
        "lidar": SparseTensor(points[inds], pc[inds]), "targets": SparseTensor(labels[inds], pc[inds]),
        "targets_mapped": SparseTensor(labels, pc), "inverse_map": SparseTensor(inverse, pc),
        "num_points": torch.tensor([points.shape[0]]),
        "lidar_ms": SparseTensor(points_ms[inds_ms], pc_ms[inds_ms]),
        "targets_ms": SparseTensor(labels_ms[inds_ms], pc_ms[inds_ms]),
        "targets_mapped_ms": SparseTensor(labels_ms, pc_ms), "inverse_map_ms": SparseTensor(inverse_ms, pc_ms),
        "num_points_ms": torch.tensor([points_ms.shape[0]]),
    }


def _stack_sparse(items: List[SparseTensor]) -> SparseTensor:
    coords = [torch.cat([t.coords, torch.full((t.coords.shape[0], 1), b, dtype=torch.int32, device=t.coords.device)], 1)
              for b, t in enumerate(items)]
    return SparseTensor(torch.cat([t.feats for t in items], 0), torch.cat(coords, 0).contiguous(), items[0].stride)


def collate_batch(samples: List[Dict]) -> Dict:
    """sparse_collate_fn + offsets + point_mask (semantickitti_voxel_ms.py:189-212), on device."""
    out = {}
    for key, first in samples[0].items():
        col = [s[key] for s in samples]
        if isinstance(first, SparseTensor):
            out[key] = _stack_sparse(col)
        elif isinstance(first, torch.Tensor):
            out[key] = torch.stack(col, 0)
        else:
            out[key] = col
    dev = out["lidar"].coords.device
    for sfx in ("", "_ms"):
        if "lidar" + sfx in out:
            sizes = torch.tensor([s["lidar" + sfx].coords.shape[0] for s in samples])
            out["offset" + sfx] = torch.cumsum(sizes, 0).int().to(dev)
    if "num_points_ms" in out:
        n_ms = [int(s["num_points_ms"]) for s in samples]
        n_cur = [int(s["num_points"]) for s in samples]
        mask = torch.zeros(sum(n_ms), dtype=torch.bool, device=dev)
        cur = 0
        for a, b in zip(n_cur, n_ms):
            mask[cur:cur + a] = True          # the current frame is the prefix of every fused cloud
            cur += b
        out["point_mask"] = mask
    return out


def build_multiscan_batch_per_sample(scans: List[Dict], voxel_size: float, steps: Sequence[int]) -> Dict:
    """build_multiscan_batch sample by sample (fuse, clamp, two voxelisations and ~55 launches per sample, then collate): the
    form the batched stage below replaced; kept as its cross-check (tests) and for TASEG_STAGE_BATCHED=0."""
    samples = []
    for s in scans:
        pts, lab, poses = s["points"], s["labels"], s["poses"]
        t = len(pts) - 1
        deltas = s.get("deltas") or [i - t for i in range(t)]
        raw_all, lab_all, keep = _fuse_history(pts[t], lab[t], pts[:t], lab[:t], poses[t], poses[:t], deltas, steps,
                                               s.get("pseudo"))
        samples.append(voxelize_sample_ms(pts[t], lab[t].long(), raw_all, lab_all, voxel_size, s.get("name", ""),
                                          keep=keep))
    return collate_batch(samples)


import os as _os

_BATCHED = options.stage_batched
_rows_cache = {}


def rows_index(lengths: Sequence[int], device) -> torch.Tensor:
    """int64 [sum(lengths)]: the index of the segment every concatenated row belongs to (built on the device from the lengths;
    a few layouts are kept - resident synthetic scans repeat theirs every step)"""
    key = (tuple(lengths), str(device))
    hit = _rows_cache.get(key)
    if hit is None:
        if len(_rows_cache) >= 64:
            _rows_cache.pop(next(iter(_rows_cache)))
        total = int(sum(lengths))
        lens = torch.tensor(list(lengths), dtype=torch.int64).to(device, non_blocking=True)
        hit = torch.repeat_interleave(torch.arange(len(lengths), dtype=torch.int64, device=device), lens, output_size=total)
        _rows_cache[key] = hit
    return hit


def voxelize_batch_ms(cur_list: List[torch.Tensor], lab_list: List[torch.Tensor], cur_ms: torch.Tensor, hist_pts: torch.Tensor,
                      hist_lab: torch.Tensor, hist_scan: torch.Tensor, hist_cls: torch.Tensor, table: torch.Tensor,
                      sample_of_scan: torch.Tensor, voxel_size: float, names: List[str], pre_keep=None, neg_col: int = -1) -> Dict:
    """collate_batch([voxelize_sample_ms(...) for every sample]) for the WHOLE batch in one chain of launches
    (semantickitti_voxel_ms.py:121-212 / nuscenes_voxel_ms.py:77-212): the per-sample clamp minima in one launch, the class-step
    rule + pre-filter + clamp in one launch (csrc/stage.hip), ONE compaction, the fused clouds written sample-major with the current
    scan first (no sort), ONE voxelisation of all fused clouds (per-sample minima, batch-keyed radix sort: voxel order (b, x, y, z),
    representative = first point, inverse map) and ONE of the current scans shifted by their fused cloud's minimum, gathers on the
    whole batch.  Four host reads per batch (kept points, kept points per sample, the two voxel counts) instead of three per
    sample.  Same tensors, bit for bit, as the per-sample path.

    cur_list[b] [n_b, F] / lab_list[b] int64: the current scans (single-frame cloud);  cur_ms [sum n_b, Fm]: the same points as
    they appear in the fused clouds (time flag / time column);  hist_*: the transformed history points of all samples, sample-major:
    points [Nh, Fm], labels [Nh] int64, global scan / sweep index [Nh] int32, pseudo class [Nh] int64 (neg_col: the table column of
    a negative class), table [S, C] bool (is class c taken from scan s), sample_of_scan [S] int64; pre_keep [Nh] bool (optional:
    the ego-box filter)."""
    dev = cur_ms.device
    nb = len(cur_list)
    if nb > 64:
        raise ValueError("voxelize_batch_ms: at most 64 samples per batch")
    n_cur = [int(c.shape[0]) for c in cur_list]
    cur = torch.cat(cur_list, 0).contiguous()
    cur_lab = torch.cat(lab_list, 0)
    cur_b = rows_index(n_cur, dev)
    n_c = cur.shape[0]
    if hist_pts.shape[0]:
        lo = B.segment_min3(cur, cur_b, nb)        # minimum of every current scan: the fused cloud is clamped to it (:121-124)
        keep, hist_b = B.stage_keep_flags(hist_pts, hist_scan, hist_cls, table, sample_of_scan, lo, pre_keep=pre_keep, neg_col=neg_col)
        idx = keep.nonzero().squeeze(1)                                 # host read 1 (the compaction's size)
        cuts = [0]
        for n in n_cur:
            cuts.append(cuts[-1] + n)
        cur_start = torch.tensor(cuts, dtype=torch.int64).to(dev, non_blocking=True)
        kept_start = torch.searchsorted(hist_b[idx], torch.arange(nb + 1, device=dev))
        ms_pts, ms_lab, ms_b, ms_b32, point_mask = B.stage_layout(cur_ms, cur_lab, cur_b, hist_pts, hist_lab, hist_b, idx, cur_start,
                                                                  kept_start)
        kept = (kept_start[1:] - kept_start[:-1]).tolist()              # host read 2 (kept history points per sample)
        n_ms = [a + k for a, k in zip(n_cur, kept)]
    else:
        ms_b, ms_b32, ms_pts, ms_lab = cur_b, cur_b.int(), cur_ms.contiguous(), cur_lab
        point_mask = torch.ones(n_c, dtype=torch.bool, device=dev)
        n_ms = list(n_cur)
    coords_ms, mins = B.voxel_coords(ms_pts, voxel_size, batch_idx=ms_b32, n_batch=nb)
    index_ms, inverse_ms = B.sparse_quantize(coords_ms)                 # host read 3 (voxels of the fused clouds)
    coords_c, _ = B.voxel_coords(cur, voxel_size, batch_idx=rows_index32(n_cur, dev), n_batch=nb, shift=mins)   # pc_ -= pc_ms_.min(0) (:130)
    index_c, inverse_c = B.sparse_quantize(coords_c)                    # host read 4 (voxels of the current scans)
    vox_ms, offset_ms, inv_ms = B.stage_split_voxels(coords_ms, index_ms, inverse_ms, ms_b, nb)
    vox_c, offset_c, inv_c = B.stage_split_voxels(coords_c, index_c, inverse_c, cur_b, nb)
    index_ms, index_c = index_ms.long(), index_c.long()
    return {
        "name": list(names),
        "lidar": SparseTensor(cur[index_c], vox_c), "targets": SparseTensor(cur_lab[index_c], vox_c),
        "targets_mapped": SparseTensor(cur_lab, coords_c), "inverse_map": SparseTensor(inv_c, coords_c),
        "num_points": torch.tensor(n_cur).view(-1, 1),
        "lidar_ms": SparseTensor(ms_pts[index_ms], vox_ms), "targets_ms": SparseTensor(ms_lab[index_ms], vox_ms),
        "targets_mapped_ms": SparseTensor(ms_lab, coords_ms), "inverse_map_ms": SparseTensor(inv_ms, coords_ms),
        "num_points_ms": torch.tensor(n_ms).view(-1, 1),
        "offset": offset_c, "offset_ms": offset_ms, "point_mask": point_mask,
    }


def rows_index32(lengths: Sequence[int], device) -> torch.Tensor:
    """rows_index as int32 (what ts_voxel_coords takes), cached beside it"""
    key = ("i32", tuple(lengths), str(device))
    hit = _rows_cache.get(key)
    if hit is None:
        hit = rows_index(lengths, device).int()
        _rows_cache[key] = hit
    return hit


def build_multiscan_batch(scans: List[Dict], voxel_size: float, steps: Sequence[int]) -> Dict:
    """scans[b] = dict(points=[T+1 tensors, current LAST], labels=[...], poses=[...], name=str
    [, deltas=[frame offsets of the history scans], pseudo=[pseudo classes of the history scans, see _fuse_history]]).
    Returns the collated batch_dict MinkUNetMs consumes.  The whole batch goes through ONE chain of launches: one pose-fuse
    launch over every history point of every sample (ts_fuse_scans_batch), the class-step rule as one table lookup, then
    voxelize_batch_ms."""
    if not _BATCHED or not scans or len(scans) > 64:
        return build_multiscan_batch_per_sample(scans, voxel_size, steps)
    dev = scans[0]["points"][-1].device
    n_cls = len(steps)
    cur_list, lab_list, hist_pts, hist_lab, hist_ps, lengths, scan_sample, pose0s, poses, rows = [], [], [], [], [], [], [], [], [], []
    for b, s in enumerate(scans):
        pts, lab, ps = s["points"], s["labels"], s["poses"]
        t = len(pts) - 1
        deltas = s.get("deltas") or [i - t for i in range(t)]
        cur_list.append(pts[t][:, :4] if pts[t].shape[1] != 4 else pts[t])
        lab_list.append(lab[t].long())
        pseudo = s.get("pseudo")
        for i in range(t):
            hist_pts.append(pts[i][:, :4])
            hist_lab.append(lab[i])
            hist_ps.append(lab[i] if pseudo is None else pseudo[i])
            lengths.append(int(pts[i].shape[0]))
            scan_sample.append(b)
            pose0s.append(ps[t])
            poses.append(ps[i])
            # semantickitti_ms.py:303-308: class c is aggregated from the scan delta frames away iff steps[c] != 0 and
            # |delta| % steps[c] == 0; last column: pseudo class -1 (no class's canonical raw id): never kept
            rows.append([bool(st) and abs(deltas[i]) % st == 0 for st in steps] + [False])
    cur4 = torch.cat(cur_list, 0)
    cur_ms = torch.cat([cur4, torch.ones((cur4.shape[0], 1), dtype=cur4.dtype, device=dev)], 1)      # append_time_flag (:253-257)
    if hist_pts:
        hp = torch.cat(hist_pts, 0).contiguous()
        hl = torch.cat(hist_lab, 0).long()
        hps = hl if all(s.get("pseudo") is None for s in scans) else torch.cat(hist_ps, 0).long()
        key = ("kitti-table", tuple(map(tuple, rows)), tuple(scan_sample), str(dev))
        hit = _static_cache.get(key)
        if hit is None:
            if len(_static_cache) >= 64:
                _static_cache.pop(next(iter(_static_cache)))
            hit = (torch.tensor(rows, dtype=torch.bool).to(dev), torch.tensor(scan_sample, dtype=torch.int64).to(dev))
            _static_cache[key] = hit
        table, sample_of_scan = hit
        scan32 = rows_index32(lengths, dev)
        fused = B.fuse_scans_batch(hp, scan32, torch.stack(pose0s, 0), torch.stack(poses, 0))
        hist_ms = torch.cat([fused, torch.zeros((fused.shape[0], 1), dtype=fused.dtype, device=dev)], 1)
    else:
        hist_ms = torch.empty((0, 5), dtype=cur4.dtype, device=dev)
        hl = hps = torch.empty(0, dtype=torch.int64, device=dev)
        scan32 = torch.empty(0, dtype=torch.int32, device=dev)
        table = torch.zeros((1, n_cls + 1), dtype=torch.bool, device=dev)
        sample_of_scan = torch.zeros(1, dtype=torch.int64, device=dev)
    return voxelize_batch_ms([s["points"][-1] for s in scans], lab_list, cur_ms, hist_ms, hl, scan32, hps, table, sample_of_scan,
                             voxel_size, [s.get("name", "") for s in scans], neg_col=n_cls)


class DevicePrefetcher:
    """Double-buffered device-side data stage: the batch of step i+1 (temporal aggregation, voxelisation and
    the model's index plan - everything that does not depend on parameters) is built on a second HIP stream
    while the launch stream still executes step i, the role the reference gives its DataLoader workers
    (tools/train.py builds the loader with workers + pin_memory so batch i+1 is ready when step i ends).

    The stage needs a few host reads (voxel counts, pair totals).  On the launch stream each of them would
    drain the whole queue of step i first; on the side stream they only wait for the stage's own kernels, so
    the host keeps running ahead of the device.

        pf = DevicePrefetcher(make_batch, model.prepare)
        for _ in range(steps):
            batch = pf.next()          # staged earlier; the current stream waits on its ready-event
            ... forward ...
            pf.prefetch_early()        # (optional) stage the following batch beside the backward pass
            ... backward / optimizer step on `batch` ...
            pf.prefetch()              # stage the following batch (optional: next() does it if needed)

    Memory: tensors of a staged batch are allocated on the side stream and consumed on the launch stream, so
    the prefetcher keeps every batch alive until an event recorded on the launch stream after its step has
    completed - nothing is returned to the side stream's pool while a kernel of the launch stream may still
    read it.
    """

    def __init__(self, make_batch, prepare=None, device=None, threaded=False, depth=1):
        self.make_batch, self.prepare = make_batch, prepare
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # depth > 1 (threaded only): that many batches staged ahead, each on a stream and a worker thread of its own - for a loop
        # whose pass is shorter than one stage (evaluation under autocast: the index plan is a chain of ~300 small launches and
        # five host reads, 2.7 ms on its own and ~4.7 ms beside a forward pass)
        self.depth = max(int(depth), 1) if threaded else 1
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self.stream = self.streams[0]
        self._turn = 0
        self._last_made = None       # depth > 1: batches are MADE one after the other (order, and make_batch need not be thread-safe)
        self._staged = []            # FIFO of (batch, ready_event) or Futures of it
        self._inflight = []          # [(batch, done_event)] handed out, possibly still read by the launch stream
        self._current = None
        # threaded=True runs the stage on a worker thread (its host reads release the GIL while they wait).  With the
        # index plan built natively without the interpreter lock (csrc/fastpath) this takes ~2 ms of host time per step
        # off the training thread: +4 ... 6 % where the step is host-bound (AMP at bs 2), nothing where it is device-bound;
        # bench.py switches it on for --amp (TASEG_STAGE_THREAD overrides).  Same batches, same bits either way.
        self._pool = None
        if threaded:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.depth, thread_name_prefix="taseg-stage")

    def _retire(self):
        if self._current is not None:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self._inflight.append((self._current, done))
            self._current = None
        live, dead = [], []
        for item in self._inflight:
            (dead if item[1].query() else live).append(item)
        self._inflight = live
        if dead and self._pool is not None:
            # a staged batch is hundreds of tensors and Python objects: letting go of it costs 0.2-0.3 ms, which the training
            # thread of a host-bound step does not have - the stage's own thread drops the last references
            self._pool.submit(dead.clear)
        del dead

    def prefetch_early(self):
        """Start staging the following batch while the CURRENT one is still in use (call it right after the forward pass has
        been issued): the stage then runs beside the backward pass instead of between two steps - what matters when the
        step is host-bound and the stage runs on a worker thread (`threaded=True`; measured: the AMP step at bs 2 waited
        2.6 ms per step for a stage that was started at the end of the previous step).  The current batch is NOT retired
        here; next() / prefetch() do that."""
        if len(self._staged) >= self.depth:
            return
        self._top_up(not self._inflight and self._current is None)

    def prefetch(self):
        if len(self._staged) >= self.depth:
            return
        self._retire()
        self._top_up(not self._inflight)

    def _top_up(self, idle):
        while len(self._staged) < self.depth:
            stream = self.streams[self._turn % self.depth]
            self._turn += 1
            if idle:
                # first use (or idle device): inputs created on the launch stream must be complete
                stream.wait_stream(torch.cuda.current_stream(self.device))
            if self._pool is None:
                self._staged.append(self._stage(stream))
                continue
            made = None
            if self.depth > 1:
                import threading
                made = threading.Event()
            self._staged.append(self._pool.submit(self._stage, stream, self._last_made, made))
            self._last_made = made

    def _stage(self, stream=None, after=None, made=None):
        stream = self.stream if stream is None else stream
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(stream), torch.no_grad():
            try:
                if after is not None:
                    after.wait()             # the batch before this one has been taken from the source
                batch = self.make_batch()
            finally:
                if made is not None:
                    made.set()
            if self.prepare is not None:
                self.prepare(batch)
            ready = torch.cuda.Event()
            ready.record(stream)
        return batch, ready

    def next(self):
        self._retire()               # the previous batch: every launch that reads it has been issued by now
        self.prefetch()
        staged = self._staged.pop(0)
        batch, ready = staged.result() if hasattr(staged, "result") else staged
        torch.cuda.current_stream(self.device).wait_event(ready)
        self._current = batch
        return batch

    def close(self):
        for staged in self._staged:
            if hasattr(staged, "result"):
                try:
                    staged.result()
                except Exception:  # noqa: BLE001 - a stage that failed has nothing left to wait for; next() reports failures
                    pass
        self._retire()
        torch.cuda.synchronize(self.device)
        self._inflight, self._staged = [], []
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
