"""Diagnostic: the device data stage alone (temporal aggregation + double voxelisation + collate), batched form against the
per-sample form, for the two multi-scan workloads of bench.py.  Wall time per batch; under `rocprofv3 --kernel-trace --stats` the
launch count and GPU time per batch follow from the totals (--reps batches per form, --only batched|per_sample).
     python tools/stage_probe.py [--reps 20] [--only batched]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd.data import nuscenes as N
from taseg_amd.data import stage as S
from taseg_amd.data.synthetic import FLEXIBLE_STEPS_KITTI, FLEXIBLE_STEPS_NUSC

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--workload", default="")
args = ap.parse_args()


def timed(fn):
    for _ in range(3):
        bd = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        bd = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / args.reps, bd


if args.workload in ("", "minkunet_ms"):
    scans, npts = bench.make_multiscans(0, 2, 120000)
    for form, fn in (("batched", S.build_multiscan_batch), ("per_sample", S.build_multiscan_batch_per_sample)):
        if args.only in ("", form):
            ms, bd = timed(lambda: fn(scans, 0.05, FLEXIBLE_STEPS_KITTI))
            print(f"minkunet_ms (4-scan TFA, bs 2) {form:10s}: {ms:6.2f} ms per batch, {npts} raw points -> "
                  f"{bd['lidar_ms'].C.shape[0]} fused voxels", flush=True)
if args.workload in ("", "nuscenes_ms"):
    samples, npts, n_sweeps = bench.make_nusc_samples(0, 4, 34700)
    for form, fn in (("batched", N.build_nuscenes_batch), ("per_sample", N.build_nuscenes_batch_per_sample)):
        if args.only in ("", form):
            ms, bd = timed(lambda: fn(samples, 0.1, FLEXIBLE_STEPS_NUSC))
            print(f"nuscenes_ms (FSA, bs 4, {n_sweeps} sweeps) {form:10s}: {ms:6.2f} ms per batch, {npts} raw points -> "
                  f"{bd['lidar_ms'].C.shape[0]} fused voxels", flush=True)
