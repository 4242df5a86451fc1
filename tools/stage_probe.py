"""Diagnostic: wall time of the device data stage (build_multiscan_batch) alone, per workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd.data.stage import build_multiscan_batch
from taseg_amd.data.synthetic import FLEXIBLE_STEPS_KITTI, FLEXIBLE_STEPS_NUSC, KITTI_TO_NUSC

for name, kw, voxel, steps in (("minkunet_ms", dict(batch=2, points=120000), 0.05, FLEXIBLE_STEPS_KITTI),
                               ("nuscenes_ms", dict(batch=4, points=34700, history=15, n_beams=32, n_az=1090,
                                                    label_map=KITTI_TO_NUSC), 0.1, FLEXIBLE_STEPS_NUSC)):
    scans, npts = bench.make_multiscans(0, **kw)
    for _ in range(3):
        bd = build_multiscan_batch(scans, voxel, steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        bd = build_multiscan_batch(scans, voxel, steps)
    torch.cuda.synchronize()
    print(f"{name}: {1e2 * (time.perf_counter() - t0):.2f} ms per batch, {npts} raw points -> "
          f"{bd['lidar_ms'].C.shape[0]} fused voxels")
