"""bench.py - scans/sec of the TASeg hot path (train step: rulebook construction + MinkUNet forward +
CE/Lovasz loss + backward + SGD step) on synthetic SemanticKITTI-shaped scans, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload minkunet|minkunet_ms]

N > 1: one rank per GPU over RCCL (the reference's launch, R/dist_train.sh:17-19 + R/train.py:247-251).  The driver
starts the ranks through torch.distributed.run; run by hand as `python bench.py --gpus N` (WORLD_SIZE unset) this
process starts the N ranks itself - before it touches the GPU - and forwards rank 0's line.  Every rank trains on its
own scans (weak scaling, gradient all-reduce overlapped with backward, SyncBN as in the reference configs).  Rank 0
prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3     # dense f32-input MFMA peak (same guide)
MFMA_F16_PEAK_TF = 2500.0    # dense f16 / bf16 MFMA peak (same guide; no sparsity)
MFMA_SPLIT_PEAK_TF = MFMA_F16_PEAK_TF / 6   # fp32 products as six bf16 MFMAs (csrc/conv_pairs_s.hip): 416.7 TF/s of fp32 flops
VOXEL = 0.05
EVENT_EVERY = 20            # per-launch HIP events bracket the conv kernels of every 20th timed step (recorded inside the
                            # block calls; an event pair costs ~3 us of device time per launch: ~2 ms on such a step,
                            # i.e. ~0.1 ms per step of the default 20-step run)


def note(msg):
    """progress line on stderr (a default run takes minutes: main measurement, CPU baseline legs, side runs)"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


class Watchdog:
    """Per-rank hang guard.  A rank that makes no progress for `timeout` seconds (a collective some peer never entered, a
    kernel that does not return) prints where it was and ends the PROCESS with exit code 3 - torch.distributed.run (or the
    parent of `bench.py --gpus N`) then ends the other ranks and the launch fails loudly instead of sitting in a collective
    until somebody's outer time limit.  beat(phase) marks progress; TASEG_BENCH_WATCHDOG_S (default 300, 0 = off)."""

    def __init__(self, rank):
        import threading
        self.timeout = float(os.environ.get("TASEG_BENCH_WATCHDOG_S", "300"))
        self.rank, self.phase, self.last = rank, "start", time.monotonic()
        self._stop = threading.Event()
        if self.timeout > 0:
            threading.Thread(target=self._run, name="bench-watchdog", daemon=True).start()

    def beat(self, phase):
        self.phase, self.last = phase, time.monotonic()

    def stop(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(min(5.0, self.timeout / 4)):
            idle = time.monotonic() - self.last
            if idle > self.timeout:
                print(f"[bench watchdog] rank {self.rank}: no progress for {idle:.0f} s in phase '{self.phase}' - giving up "
                      f"(exit code 3); a peer rank stuck in another collective or a kernel that never returned",
                      file=sys.stderr, flush=True)
                os._exit(3)


def host_cores():
    """CPUs this process may really use: scheduler affinity capped by the cgroup CPU quota (a GPU box hands out a
    share of a large host - running one OpenMP thread per visible core of the HOST would oversubscribe the share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read().split()[0])
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="minkunet", choices=["minkunet", "minkunet_ms", "nuscenes_ms", "tiaf", "kd"],
                    help="minkunet = BASELINE configs[1] (headline); minkunet_ms = configs[2] (4-scan TFA); "
                         "nuscenes_ms = configs[4] shape: 32-beam 34.7k-point sweeps, 15 history sweeps, voxel 0.1 m, "
                         "17 classes, bs 4 (fp32 here); tiaf = MinkUNetMsMm (temporal image aggregation and fusion, "
                         "minkunet_mk34_cr10_fsa_tiaf.yaml): 16 fused history scans, 5 camera frames of 384 x 1280 per sample, bs 2; "
                         "kd = MinkUNetMsKd (mask distillation, minkunet_mk34_cr10_fsa_kd.yaml): 16 history scans fused twice per sample "
                         "(pseudo-label masks for the student, ground-truth masks for the frozen teacher), bs 6")
    ap.add_argument("--batch", type=int, default=None, help="scans per GPU per step (default 2; 4 for nuscenes_ms)")
    ap.add_argument("--points", type=int, default=None, help="points per scan (default 120000; 34700 for nuscenes_ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="build batch i's rulebooks at the head of step i on the launch stream instead of "
                         "during step i-1 on the staging stream")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline = null)")
    ap.add_argument("--amp", action="store_true",
                    help="torch.autocast(float16) + loss scaling, the reference's default training mode (dist_train.sh "
                         "--amp): half-storage conv / BN kernels with fp32 accumulation; reported as dtype f16")
    ap.add_argument("--local-bn", action="store_true",
                    help="plain BatchNorm (per-rank statistics) instead of the reference configs' SyncBatchNorm")
    ap.add_argument("--torch-optim", action="store_true",
                    help="torch.optim.SGD + clip_grad_norm_ (+ GradScaler under --amp) instead of taseg_amd.optim.FlatSGD")
    ap.add_argument("--torch-ddp", action="store_true",
                    help="average gradients with torch's DistributedDataParallel instead of parallel.GradBucketReducer")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-GPU code path (RCCL process group, DDP, SyncBatchNorm collectives) even with "
                         "one rank - a single-GPU check of what `--gpus N` executes")
    ap.add_argument("--conv-impl", type=int, default=0,
                    help="ts_set_conv_impl: 0 = default (full-tile fp32 GEMMs as split-bf16 MFMAs), 5 = v_mfma_f32_16x16x4_f32")
    ap.add_argument("--cpu-sector-deg", type=float, default=360.0,
                    help="azimuth sector of one scan the 1-thread CPU baseline leg runs on (360 = the whole scan, ~15 s per pass)")
    ap.add_argument("--cpu-passes", type=int, default=3, help="timed passes of the 1-thread CPU leg (value = their median)")
    ap.add_argument("--cpu-warmup", type=int, default=1, help="un-timed passes of the 1-thread CPU leg")
    ap.add_argument("--no-wgrad-tune", action="store_true",
                    help="skip the timing of the weight gradients on a second stream (TASEG_WGRAD_STREAM=0 / 1 pins the setting)")
    ap.add_argument("--cpu-sector-deg-all", type=float, default=45.0,
                    help="sector of the all-cores leg (the reference's CPU convolution gets SLOWER with threads: its "
                         "OpenMP pragma is on the inner channel loop; 0 = skip the leg)")
    ap.add_argument("--cpu-leg", type=int, default=0,
                    help="(internal) run ONE CPU-baseline leg with this many threads, print its JSON record and exit")
    ap.add_argument("--cpu-leg-timeout", type=float, default=240.0, help="seconds before the 1-thread CPU-baseline leg is abandoned")
    ap.add_argument("--eval", action="store_true",
                    help="evaluation pass instead of a training step: eval-mode forward on the index plan + un-voxelisation + "
                         "arg-max per point, as the segmentors' eval branch returns it (minkunet.py:435-455, R/train.py:452-540)")
    ap.add_argument("--history", type=int, default=None,
                    help="minkunet_ms: history scans fused per sample (default 4 = BASELINE configs[2]; the reference's FSA recipe, "
                         "minkunet_mk34_cr10_fsa.yaml:14,34, is 16 at bs 6)")
    ap.add_argument("--image-layout", default=None, choices=["nhwc", "nchw"],
                    help="tiaf: memory format of UNet2D and form of the image gather (default: taseg_amd.options.image_layout = nhwc)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short runs of the other workloads (minkunet_ms, --amp, nuscenes_ms --amp) that the "
                         "default N=1 run appends as `secondary`")
    return ap.parse_args()


def make_scans(rank, batch, points, workload):
    """Seeded synthetic scans (seed = 1000 * seq + frame, seq = rank), dataset-voxelised on the host with
    numpy exactly like the reference's CPU workers, returned as resident device tensors."""
    from taseg_amd.data.synthetic import synth_scan
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    coords, feats, labels, npts = [], [], [], 0
    for b in range(batch):
        seed = 1000 * rank + 10 * b
        pts, lab = synth_scan(seed, n_points=points)
        pc = np.round(pts[:, :3] / VOXEL).astype(np.int32)
        pc -= pc.min(0, keepdims=True)
        _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
        coords.append(np.concatenate([pc[idx], np.full((len(idx), 1), b, np.int32)], 1))
        feats.append(pts[idx])
        labels.append(lab[idx].astype(np.int64))
        npts += len(pts)
    dev = torch.device("cuda")
    return (torch.from_numpy(np.concatenate(coords)).to(dev), torch.from_numpy(np.concatenate(feats)).to(dev),
            torch.from_numpy(np.concatenate(labels)).to(dev), npts)


def _parallel_scans(jobs):
    """[(seed, synth_scan keyword arguments)] -> [(points, labels)] in order, on up to 8 host threads"""
    from concurrent.futures import ThreadPoolExecutor
    from taseg_amd.data.synthetic import synth_scan
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        return list(ex.map(lambda job: synth_scan(job[0], **job[1]), jobs))


def make_multiscans(rank, batch, points, history=4, n_beams=64, n_az=2083, label_map=None, pseudo_flip=0.0):
    """Raw resident scans for the "4-scan TFA" workload: per sample the
    current scan plus `history` earlier scans of the same scene seen from the ego poses
    synth_pose(t) (1.1 m and 0.4 deg per frame).  The temporal aggregation + voxelisation itself runs on the
    device INSIDE the timed step (taseg_amd.data.stage.build_multiscan_batch)."""
    from taseg_amd.data.synthetic import synth_pose, synth_scan
    dev = torch.device("cuda")
    scans, npts = [], 0
    # (the synthetic scans are numpy work of ~1 s each - 102 of them for the FSA recipe at bs 6: generated on a few host threads, every
    # scan from its own seed, so the clouds do not depend on the order they are made in)
    made = _parallel_scans([(1000 * rank + 10 * b + t, dict(n_points=points, pose=synth_pose(history - t), scene_seed=1000 * rank + 10 * b,
                                                            n_beams=n_beams, n_az=n_az)) for b in range(batch) for t in range(history + 1)])
    for b in range(batch):
        seed = 1000 * rank + 10 * b
        pts, labs, poses = [], [], []
        for t in range(history + 1):               # frame t: current = history, oldest = 0
            pose = synth_pose(history - t)
            p, l = made[b * (history + 1) + t]
            if label_map is not None:
                l = np.asarray(label_map, dtype=l.dtype)[l]
            pts.append(torch.from_numpy(p).to(dev))
            labs.append(torch.from_numpy(l.astype(np.int64)).to(dev))
            poses.append(torch.from_numpy(pose).to(dev))
            npts += len(p)
        scans.append({"points": pts, "labels": labs, "poses": poses, "name": f"{rank}/{b}"})
        if pseudo_flip > 0:
            # pseudo labels of the history scans (the reference reads a trained model's predictions, semantickitti_ms_kd.py:318-323):
            # the annotation with a share of the points moved to another class
            g = torch.Generator(device="cpu").manual_seed(seed + 7)
            ps = []
            for l in labs[:-1]:
                flip = (torch.rand(l.shape[0], generator=g) < pseudo_flip).to(dev)
                other = torch.randint(1, 20, (l.shape[0],), generator=g).to(dev)
                ps.append(torch.where(flip, other, l))
            scans[-1]["pseudo"] = ps
    return scans, npts


TIAF_HEIGHT, TIAF_WIDTH, TIAF_MULTISCAN, TIAF_MULTISCAN_IMAGE, TIAF_STEP_IMAGE = 384, 1280, 16, 48, 12   # tiaf yaml :17-22


def make_tiaf_frames(rank, batch, points, n_beams=64, n_az=2083):
    """Resident inputs of the TIAF workload (R/tools/cfgs/voxel/semantic_kitti/minkunet_mk34_cr10_fsa_tiaf.yaml: MULTISCAN 16,
    MULTISCAN_IMAGE 48, STEP_IMAGE 12, HEIGHT 384, WIDTH 1280, bs 2): per sample the current scan, the 16 history scans the LiDAR
    aggregation walks and the scans of the camera frames at -24 / -36 / -48, each with labels, pseudo labels and pose; a
    KITTI-sized uint8 camera image (376 x 1241) + label map for the current frame and every 12th history frame; P2 @ Tr of a
    camera looking along +x.  Everything the reference's dataset does with them (projection, FOV test, crop / pad, temporal
    aggregation, three-cloud voxelisation, collate: semantickitti_ms_mm.py:304-461, semantickitti_voxel_ms_mm.py:79-266) runs on
    the device INSIDE the timed step (taseg_amd.data.tiaf)."""
    from taseg_amd.data.synthetic import synth_pose, synth_scan
    dev = torch.device("cuda")
    with_img = [d for d in range(0, -TIAF_MULTISCAN_IMAGE - 1, -TIAF_STEP_IMAGE)]
    deltas = sorted(set(range(0, -TIAF_MULTISCAN - 1, -1)) | set(with_img))
    proj = torch.tensor([[609.5, -721.5, 0.0, -44.86], [172.85, 0.0, -721.5, -216.4], [1.0, 0.0, 0.0, -0.27]], dtype=torch.float64, device=dev)
    samples, npts = [], 0
    for b in range(batch):
        seed = 1000 * rank + 100 * b
        rs = np.random.RandomState(seed)
        frames = {}
        made = _parallel_scans([(seed - d, dict(n_points=points, pose=synth_pose(-d), scene_seed=seed, n_beams=n_beams, n_az=n_az)) for d in deltas])
        for d, (p, l) in zip(deltas, made):
            pose = synth_pose(-d)
            lab = torch.from_numpy(l.astype(np.int64)).to(dev)
            f = {"points": torch.from_numpy(p).to(dev), "labels": lab, "pseudo": lab, "pose": torch.from_numpy(pose).to(dev)}
            if d in with_img:
                f["image"] = torch.from_numpy(rs.randint(0, 256, (376, 1241, 3), dtype=np.uint8)).to(dev)
                f["semantic"] = torch.from_numpy(rs.randint(0, 20, (376, 1241, 1)).astype(np.float32)).to(dev)
            frames[d] = f
            npts += len(p)
        samples.append(frames)
    return samples, proj, npts


def tiaf_gather_roofline(model, bd, layout="nhwc", dtype=torch.float32):
    """The image -> point gather and its adjoint (csrc/image.hip) alone on the device, at the shapes of this batch, in the memory
    format and element type the step uses (channels-last rows / NCHW planes; fp32 / the fp16 maps of an autocast step): HIP events
    around 20 launches each on the current stream; algorithmic bytes n * C * s gathered + n * C * s written (forward), n * C * s
    read + 2 * s * C per touched pixel read-modify-written (adjoint, into the map's other gradient), s = element size."""
    from taseg_amd import backend as B
    rows = layout == "nhwc"
    es = 2 if (rows and dtype == torch.float16) else 4
    fov = bd["lidar_fov_ms"]
    pix = fov.F[:, -2:].float().contiguous()
    pbatch = fov.C[:, -1].int().contiguous()
    frame_end = bd["offset_img"].int().contiguous()
    T = int(bd["image_ms"].shape[0])
    out = []
    for name, c, shift in (("u4 (full scale, 96 channels)", 96, 0), ("u2 (1/4 scale, 128 channels)", 128, 2), ("logits (full scale, 20 channels)", 20, 0)):
        plan = B.image_plan(pix, pbatch, frame_end, T, TIAF_HEIGHT, TIAF_WIDTH, shift)
        hs, ws = plan["shape"]
        feat = torch.randn(T, c, hs, ws, device="cuda")
        gout = torch.randn(plan["n"], c, device="cuda")
        if rows:
            feat = feat.to(dtype).contiguous(memory_format=torch.channels_last)
            gout = gout.to(dtype)
            fwd, adj = (lambda: B.image_gather_rows_forward(feat, plan)), (lambda: B.image_gather_rows_backward(gout, plan, c, into=feat))
        else:
            fwd, adj = (lambda: B.image_gather_forward(feat, plan)), (lambda: B.image_gather_backward(gout, plan, c, into=feat))
        touched = int((plan["run"][:plan["n"]] > 0).sum())
        rec = {"map": name, "points": plan["n"], "touched_pixels": touched, "layout": layout, "element_bytes": es}
        for kind, fn, byts in (("forward", fwd, plan["n"] * c * 2.0 * es),
                               ("adjoint", adj, plan["n"] * c * 1.0 * es + touched * c * 2.0 * es)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            e1.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 20
            rec[kind] = {"avg_us": us, "algorithmic_bytes": byts, "achieved_GBs": byts / us / 1e3, "hbm_frac": byts / us / 1e3 / HBM_PEAK_GBS}
        out.append(rec)
        del feat, gout
    return out


def tiaf_phase_table(model, net, opt, make_batch, amp, steps=5):
    """Where a TIAF step's DEVICE time goes, by branch: `steps` further steps after the timed region, staged in line (no
    prefetcher), with HIP events at the branch boundaries - forward hooks around UNet2D (with its gathers) and UNet3D, a tensor hook
    on the gathered image features (their gradient is complete when the sparse branches' backward is: what follows is the gather
    adjoints + UNet2D's backward; the dense image loss's own backward runs first and is counted with the sparse part).
    Milliseconds per step, the MEDIAN over the steps (a step of this in-line form now and then carries an allocator growth or a host
    hiccup inside one phase: with three steps and means one of them moved a phase by 10 ms)."""
    marks_all = []
    cur = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        cur[name] = e

    def tensor_mark(name):
        def hook(grad):
            if name not in cur:
                mark(name)
            return grad
        return hook

    def after_unet2d(mod, inp, out):
        mark("unet2d_out")
        feats = out["image_features_fov"]
        if feats.requires_grad:
            feats.register_hook(tensor_mark("bwd_features"))

    hooks = [model.image_backbone.register_forward_pre_hook(lambda m, i: mark("unet2d_in")),
             model.image_backbone.register_forward_hook(after_unet2d),
             model.lidar_backbone.register_forward_pre_hook(lambda m, i: mark("unet3d_in")),
             model.lidar_backbone.register_forward_hook(lambda m, i, o: mark("unet3d_out"))]
    try:
        for _ in range(steps):
            cur = {}
            opt.zero_grad(set_to_none=True)
            mark("start")
            bd = make_batch()
            model.prepare(bd)
            mark("staged")
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                ret, _, _ = net(bd)
            loss = ret["loss"].float().mean()
            mark("forward_done")
            (loss * opt.loss_scale()).backward()
            mark("backward_done")
            opt.step()
            mark("optimizer_done")
            marks_all.append(cur)
        torch.cuda.synchronize()
    finally:
        for h in hooks:
            h.remove()
    spans = (("data stage (16-scan fuse x 3 clouds, projection, voxelisation, index plan)", "start", "staged"),
             ("forward: up to UNet2D (input stack)", "staged", "unet2d_in"),
             ("forward: UNet2D + image gathers", "unet2d_in", "unet2d_out"),
             ("forward: FOV feature concat", "unet2d_out", "unet3d_in"),
             ("forward: UNet3D (FOV cloud)", "unet3d_in", "unet3d_out"),
             ("forward: MinkUNet on the fused cloud + fusion head + five losses", "unet3d_out", "forward_done"),
             ("backward: losses, fusion head, MinkUNet, UNet3D, dense image loss", "forward_done", "bwd_features"),
             ("backward: gather adjoints + UNet2D", "bwd_features", "backward_done"),
             ("optimizer (clip + SGD)", "backward_done", "optimizer_done"))
    out = {}
    for name, a, b in spans:
        vals = sorted(m[a].elapsed_time(m[b]) for m in marks_all if a in m and b in m)
        out[name] = round(vals[len(vals) // 2], 3) if vals else None
    whole = sorted(m["start"].elapsed_time(m["optimizer_done"]) for m in marks_all)
    out["step (staged in line)"] = round(whole[len(whole) // 2], 3)
    return out


def miopen_find_db():
    """UNet2D's dense 2-D convolutions run on MIOpen (out of the hand-written scope).  On a machine that has never
    run these shapes MIOpen first measures its solvers and compiles the chosen kernels: ~3 minutes before the first TIAF step, and a
    line is only as good as the solvers that search ends with.  taseg_amd/data/miopen_gfx950/ holds what that search produced for the
    shapes of the TIAF workload on this ROCm (MIOpen's user find-db, text, + its kernel cache): copied into a private directory that
    MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR point at (MIOpen writes there), unless the caller has set either variable.  A
    db of another MIOpen build / device is ignored by MIOpen itself (file names carry version, arch and CU count)."""
    if os.environ.get("MIOPEN_USER_DB_PATH") or os.environ.get("MIOPEN_CUSTOM_CACHE_DIR") or os.environ.get("TASEG_BENCH_MIOPEN_DB") == "0":
        return None
    src = os.path.join(ROOT, "taseg_amd", "data", "miopen_gfx950")
    if not os.path.isdir(src):
        return None
    import shutil
    import tempfile
    dst = tempfile.mkdtemp(prefix="taseg_miopen_")
    for f in os.listdir(src):
        if f.endswith((".txt", ".ukdb")):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = dst
    return dst


def make_nusc_samples(rank, batch, points, multiscan=15, step=1.0):
    """Resident inputs of the nuScenes FSA workload (BASELINE configs[4]): per sample the current keyframe of a synthetic
    20 Hz drive plus the sweeps the reference's distance rule selects (nuscenes_ms.py:238-276: the frame nearest to every
    metre of driven distance up to 15 m, and every keyframe on the way), their pseudo labels, and the per-sweep
    transform parameters.  Sweep selection and the 3x3 pose products are host work the reference caches per keyframe
    (`token2samplelist`); ego-box filter, transforms, class-step mask and voxelisation run on the device inside the
    timed step (taseg_amd.data.nuscenes.build_nuscenes_batch)."""
    from taseg_amd.data.nuscenes import select_sweeps, sweep_params
    from taseg_amd.data.synthetic import KITTI_TO_NUSC, synth_nusc_sequence, synth_scan
    dev = torch.device("cuda")
    seq, world = synth_nusc_sequence()
    index = len(seq.global_indexes) - 1
    g0 = int(seq.global_indexes[index])
    offsets = select_sweeps(seq, index, multiscan, step)
    params = torch.from_numpy(sweep_params(seq, index, offsets)).to(dev)
    remap = np.asarray(KITTI_TO_NUSC, dtype=np.int64)
    samples, npts = [], 0

    def frame(seed, g):
        p, l = synth_scan(seed + g, n_points=points, n_beams=32, n_az=1090, pose=world[g].astype(np.float32), scene_seed=seed)
        p5 = np.concatenate([p, np.zeros((len(p), 1), np.float32)], 1)       # x, y, z, intensity, ring
        return torch.from_numpy(p5).to(dev), torch.from_numpy(remap[l]).to(dev)

    for b in range(batch):
        seed = 1000 * rank + 100 * b
        cur, cur_lab = frame(seed, g0)
        hp, hl, hs = [], [], []
        for d in offsets:
            p5, lab = frame(seed, g0 + d)
            hp.append(p5)
            hs.append(lab)                                                    # pseudo labels = the synthetic labels
            hl.append(lab if seq.is_key[g0 + d] else torch.zeros_like(lab))   # sweeps carry no annotation (:318)
            npts += len(p5)
        npts += len(cur)
        samples.append(dict(points=cur, labels=cur_lab, hist_points=hp, hist_labels=hl, hist_pseudo=hs, params=params,
                            name=f"{rank}/{b}"))
    return samples, npts, len(offsets)


def _cpu_leg(cfg, points, sector_deg, threads, passes=1, warmup=0):
    """Timed passes (forward + CE/Lovasz loss + backward) of the reference CPU path over scan seed 0 (an azimuth sector of it for
    sector_deg < 360) with `threads` OpenMP / BLAS threads: the oracle model driven by the reference's own compiled CPU kernels
    (oracle/_ref) when present, else the numpy port.  value = scans per second at the MEDIAN timed pass."""
    from oracle import model as OM
    from taseg_amd.data.synthetic import fill_parameters, synth_scan
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    torch.set_num_threads(threads)
    kind = "reference"
    try:
        OM._load_ref()
    except Exception:
        kind = "port"
    pts, lab = synth_scan(0, n_points=points)
    az = np.degrees(np.arctan2(pts[:, 1], pts[:, 0]))
    keep = np.abs(az) <= sector_deg / 2
    pts, lab = pts[keep], lab[keep]
    pc = np.round(pts[:, :3] / VOXEL).astype(np.int32)
    pc -= pc.min(0, keepdims=True)
    _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
    coords = np.concatenate([pc[idx], np.zeros((len(idx), 1), np.int32)], 1)
    feats = torch.from_numpy(pts[idx])
    if cfg.IN_FEATURE_DIM == 5:
        feats = torch.cat([feats, torch.ones(len(feats), 1)], 1)      # time flag of a current-frame point
    labels = torch.from_numpy(lab[idx].astype(np.int64))
    model = fill_parameters(build_network(cfg, 20), seed=0)
    learn = {k for k, _ in model.named_parameters()}
    params = {k: v.clone().requires_grad_(k in learn) for k, v in model.state_dict().items()}
    om = OM.OracleMinkUNet(params, cfg, backend="ref" if kind == "reference" else "numpy", training=True)
    fwd = om.forward_minkunet if cfg.NAME == "MinkUNet" else om.forward_minkunet_ms
    times = []
    for i in range(warmup + passes):          # 1 warm-up, median of >= 3 timed steps
        for v in params.values():
            v.grad = None
        t0 = time.time()
        logits = fwd(coords, feats)
        loss = OM.loss_ce_lovasz(logits, labels)
        loss.backward()
        if i >= warmup:
            times.append(time.time() - t0)
    dt = float(np.median(times))
    frac = float(keep.sum()) / float(points)
    what = "one whole scan" if sector_deg >= 360 else f"{sector_deg:g} deg azimuth sector of one scan"
    return {"value": frac / dt, "unit": "scans/s", "threads": threads, "kind": kind, "passes_s": [round(t, 2) for t in times],
            "sample": f"{what}: {int(keep.sum())} pts -> {len(idx)} voxels ({frac:.3f} scan), fwd+bwd median {dt:.1f} s of "
                      f"{passes} timed pass(es) after {warmup} warm-up, fp32, {threads} thread(s)"}


def cpu_baseline(args, cfg_name, in_dim):
    """Reference CPU path on bounded samples, at 1 thread (the reference's fastest setting: its OpenMP pragma sits on
    the inner channel loop) AND at all host cores; the headline `value` is
    the faster of the two, `cores` the threads that leg used.  Each leg is a child process (a fresh OpenMP runtime with
    its own thread count) under a time limit: with one thread per core the reference's convolution can be orders of
    magnitude slower than with one."""
    import subprocess
    nproc = host_cores()
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    legs = []
    # 1 thread: a WHOLE scan, 1 warm-up pass + the median of 3 timed passes; all cores: one un-warmed pass
    # over a 45 deg sector - with one thread per core the reference's convolution is ~20x slower and a whole scan would not finish
    # inside the bench's time budget
    plan = [(1, args.cpu_sector_deg, args.cpu_passes, args.cpu_warmup, args.cpu_leg_timeout)] + \
           ([(nproc, args.cpu_sector_deg_all, 1, 0, 100.0)] if nproc > 1 and args.cpu_sector_deg_all > 0 else [])
    for threads, sector, passes, warm, limit in plan:
        note(f"cpu_baseline leg: {threads} thread(s), {sector:g} deg sector, {warm} warm-up + {passes} timed pass(es)")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", str(threads), "--cpu-sector-deg", str(sector),
               "--cpu-passes", str(passes), "--cpu-warmup", str(warm), "--points", str(args.points), "--workload", args.workload]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
        rec = {"threads": threads, "value": None, "unit": "scans/s"}
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and line:
                rec = json.loads(line[-1])
            else:
                rec["error"] = (r.stderr or r.stdout)[-300:]
        except subprocess.TimeoutExpired:
            rec["sample"] = (f"{sector:g} deg azimuth sector of one scan did not finish fwd+bwd within "
                             f"{limit:g} s at {threads} threads")
        legs.append(rec)
    done = [r for r in legs if r.get("value")]
    if not done:
        return {"value": None, "unit": "scans/s", "cores": None, "kind": None, "nproc": nproc, "cpu_model": cpu_model,
                "legs": legs}
    best = max(done, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "scans/s", "cores": best["threads"], "kind": best["kind"],
            "sample": best["sample"], "nproc": nproc, "cpu_model": cpu_model, "legs": legs}


def summarise_profile(records, steps):
    """Per kernel instantiation: launches, mean duration, algorithmic flops / bytes (SURVEY.md section 8(d):
    flops = 2 P Cin Cout per pass; bytes of a forward / input-gradient pass = P (Cin s + 2 Cout s + 8) + K Cin Cout s,
    s = 4, split over its two launches; a weight gradient reads P (Cin + Cout) s + 8 P and writes K Cin Cout 4)."""
    pairs_cache = {}
    groups = {}
    for kind, e0, e1, m in records:
        ms = e0.elapsed_time(e1)
        if kind == "conv_nbr":
            key = m["nbr"].data_ptr()
            if key not in pairs_cache:
                pairs_cache[key] = int((m["nbr"] >= 0).sum())
            p = pairs_cache[key]
        elif kind == "conv_wgrad" and "nboffs" in m:
            key = m["nboffs"].data_ptr()
            if key not in pairs_cache:
                pairs_cache[key] = int(m["nboffs"][-1])
            p = pairs_cache[key]
        else:
            p = m["pairs"]
        flops = 2.0 * p * m["c_red"] * m["c_out"]
        es = m.get("esize", 4)       # bytes per stored feature element (2 on the half-storage path)
        if kind == "pair_gemm":      # gather read + Z write + rulebook + weights  (first half of the 8(d) figure)
            byts = p * (m["c_red"] * es + m["c_out"] * es + 8) + m["k"] * m["c_red"] * m["c_out"] * es
        elif kind == "class_gemm":   # gather read + Z' write (<= 3 rows per output; direct plans: the result rows) + the plan's
            # neighbour table + weights
            byts = (p * m["c_red"] * es + m["z_rows"] * (m["c_out"] * es + 36) + m.get("out_rows", 0) * (m["c_out"] * es + 4 * 8)
                    + m["k"] * m["c_red"] * m["c_out"] * es)
        elif kind == "gather_sum":   # Z read + output write + position table       (second half)
            flops = float(p * m["c_out"])
            byts = p * m["c_out"] * es + m["n_rows"] * m["c_out"] * es + m["k"] * m["n_rows"] * 4 + m.get("side_bytes", 0.0)
        elif kind == "conv_wgrad":   # both gathered operands + rulebook + the gradient written once (no scatter term)
            byts = p * (m["c_red"] * es + m["c_out"] * es + 8) + m["k"] * m["c_red"] * m["c_out"] * 4
        elif kind == "wgrad_reduce": # partial tiles read once + the gradient written (csrc/block.hip, second-stream form)
            byts = m["bytes"]
        else:                        # single-launch forms (conv_nbr, conv_os): gather read + output rows + table + weights
            byts = p * (m["c_red"] * es + 8) + m.get("n_out", m.get("n_rows", 0)) * m["c_out"] * es + m["k"] * m["c_red"] * m["c_out"] * es
        # ideal-fused lower bound of the same launch: every feature row read / written once, the
        # rulebook and the weights once - no per-pair traffic (pair GEMM + gather-sum together make one convolution pass)
        rows = m.get("n_rows", 0)
        if kind in ("pair_gemm", "class_gemm"):
            ideal = rows * m["c_red"] * es + 8 * p + m["k"] * m["c_red"] * m["c_out"] * es
        elif kind == "gather_sum":
            ideal = rows * m["c_out"] * es
        elif kind == "conv_wgrad":
            ideal = rows * m["c_red"] * es + m.get("n_rows_b", 0) * m["c_out"] * es + 8 * p + m["k"] * m["c_red"] * m["c_out"] * 4
        elif kind == "wgrad_reduce":
            ideal = 0.0                  # (a fused weight gradient writes its result once: counted with conv_wgrad)
        else:
            ideal = byts
        g = groups.setdefault(m["name"], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "ideal": 0.0})
        g["launches"] += 1
        g["ms"] += ms
        g["flops"] += flops
        g["bytes"] += byts
        g["ideal"] += ideal
    out = []
    for name, g in groups.items():
        sec = g["ms"] / 1e3
        out.append({"kernel": name, "launches_per_step": g["launches"] / steps, "avg_us": 1e3 * g["ms"] / g["launches"],
                    "ms_per_step": g["ms"] / steps, "tflops": g["flops"] / sec / 1e12, "gbs": g["bytes"] / sec / 1e9,
                    "flops_per_launch": g["flops"] / g["launches"], "bytes_per_launch": g["bytes"] / g["launches"],
                    "bytes_per_step": g["bytes"] / steps, "ideal_fused_bytes_per_step": g["ideal"] / steps})
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


FAMILIES = (("class_gemm", "class-sorted implicit GEMM (pass 1 with the sums of a z-plane of offsets in the accumulators)"),
            ("pair_gemm", "pair GEMM (pass 1: gather -> per-offset GEMM -> Z)"),
            ("gather", "gather-sum (pass 2: Z rows -> output rows)"),
            ("wgrad", "weight gradient"),
            ("conv_os", "output-stationary fused convolution"),
            ("conv_nbr", "neighbour-table convolution"))


def family_of(kernel):
    for prefix, _ in FAMILIES:
        if kernel.startswith(prefix):
            return prefix
    return kernel.split("<", 1)[0]


def mfma_peak_of(kernel, amp):
    """dense peak the instantiation's matrix instructions are priced against (MI355X_MICROARCH.md): f16 MFMA for the
    half-storage kernels, 2500 / 6 for fp32 products as six bf16 MFMAs, the f32 MFMA otherwise"""
    if amp or "_h_kernel" in kernel:
        return MFMA_F16_PEAK_TF
    if any(t in kernel for t in ("_s_kernel", "_d_kernel", "conv_os", "class_gemm")):
        return MFMA_SPLIT_PEAK_TF
    return MFMA_F32_PEAK_TF


def summarise_families(prof, amp, traffic_table):
    """Instantiations grouped by kernel family: time share, launches, algorithmic flops and bytes, BOTH roofline
    fractions (matrix pipe and HBM), PMC traffic where every member has a traffic.json entry."""
    fams = {}
    for r in prof:
        f = fams.setdefault(family_of(r["kernel"]), {"ms_per_step": 0.0, "launches_per_step": 0.0, "flops": 0.0, "bytes": 0.0,
                                                     "ideal": 0.0, "t_mfma": 0.0, "members": [], "traffic": 0.0, "traffic_ok": True})
        f["ms_per_step"] += r["ms_per_step"]
        f["launches_per_step"] += r["launches_per_step"]
        f["flops"] += r["flops_per_launch"] * r["launches_per_step"]
        f["bytes"] += r["bytes_per_step"]
        f["ideal"] += r["ideal_fused_bytes_per_step"]
        f["t_mfma"] += r["flops_per_launch"] * r["launches_per_step"] / (mfma_peak_of(r["kernel"], amp) * 1e12)
        f["members"].append(r["kernel"])
        entry = traffic_table.get(r["kernel"])
        if entry is None:        # rocprofv3 prints every template argument (the trailing diagnostics flag too): match by prefix
            stem = r["kernel"][:-1] if r["kernel"].endswith(">") else r["kernel"]
            hits = [v for k, v in traffic_table.items() if k.startswith(stem + ",") or k == stem + ">"]
            if hits:             # several instantiations behind one label (forward / transposed): weighted by the launches sampled
                w = [max(1, h.get("launches_sampled", 1)) for h in hits]
                entry = {"hbm_bytes_per_launch": sum(h["hbm_bytes_per_launch"] * x for h, x in zip(hits, w)) / sum(w)}
        if entry is None:
            f["traffic_ok"] = False
        else:
            f["traffic"] += entry["hbm_bytes_per_launch"] * r["launches_per_step"]
    out = []
    for name, f in fams.items():
        sec = f["ms_per_step"] / 1e3
        gemm = name not in ("gather",)
        tf = f["flops"] / sec / 1e12
        gbs = f["bytes"] / sec / 1e9
        peak_tf = f["flops"] / f["t_mfma"] / 1e12 if f["t_mfma"] > 0 else None      # launch-weighted peak of the members
        out.append({"family": name, "ms_per_step": f["ms_per_step"], "launches_per_step": f["launches_per_step"],
                    "avg_us": 1e3 * f["ms_per_step"] / f["launches_per_step"],
                    "flops_per_step": f["flops"], "bytes_per_step": f["bytes"], "ideal_fused_bytes_per_step": f["ideal"],
                    "tflops": tf if gemm else None, "mfma_peak_tflops": peak_tf if gemm else None,
                    "mfma_frac": (tf / peak_tf) if gemm and peak_tf else None,
                    "gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                    "traffic_bytes_per_launch": (f["traffic"] / f["launches_per_step"]) if f["traffic_ok"] else None,
                    "kernels": f["members"]})
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def small_layer_share(records, steps, rows_below=64000):
    """the convolution launches of the deep levels (strides 8 / 16 at bs 2: fewer than `rows_below` rows of the gathered matrix) as an
    entry of their own: launches and milliseconds per step - < 10 % of the flops in ~40 % of the convolution launches"""
    ms = n = all_ms = all_n = 0.0
    for kind, e0, e1, m in records:
        t = e0.elapsed_time(e1)
        all_ms += t
        all_n += 1
        rows = m.get("n_rows") or m.get("n_out") or 0
        if 0 < rows < rows_below:
            ms += t
            n += 1
    if not all_n:
        return None
    return {"rows_below": rows_below, "launches_per_step": n / steps, "ms_per_step": ms / steps, "avg_us": 1e3 * ms / max(n, 1),
            "share_of_conv_launches": n / all_n, "share_of_conv_ms": ms / all_ms}


def launches_of(args):
    """kernel launches per step of this configuration on all streams, from the committed rocprofv3 kernel trace of the tree
    (profiles/launches.json: written by tools/collect_profiles.sh; None when the configuration has no entry)"""
    path = os.path.join(ROOT, "profiles", "launches.json")
    if not os.path.exists(path):
        return None
    default_bs = {"nuscenes_ms": 4, "kd": 6}.get(args.workload, 2)
    key = ("eval " if getattr(args, "eval", False) else "") + args.workload + (" amp" if args.amp else "") + \
        (" force-dist" if args.force_dist else "") + (f" bs{args.batch}" if args.batch not in (None, default_bs) else "") + \
        (f" history{args.history}" if getattr(args, "history", None) else "")
    return json.load(open(path)).get(key)


def build_roofline(prof, amp, bracket_us):
    """The `roofline` object of the JSON line: the kernel FAMILY with the largest share of the step's convolution time
    (all its instantiations together), bound = the larger of its matrix-pipe and HBM times at peak, both fractions stated;
    `families` lists the same for every family; `whole_step_lower_bound_ms` = max(ideal-fused bytes / HBM peak,
    GEMM flops / matrix peak) of the step's convolution work."""
    if not prof:
        return None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(tpath)) if os.path.exists(tpath) else {}
    fams = summarise_families(prof, amp, table)
    # rocprofv3-equivalent figures next to the event-bracketed ones: an event pair spans the launch gap (`event_bracket_us`, measured
    # with nothing between the two records); avg_us_net = avg_us - event_bracket_us is what `rocprofv3 --kernel-trace --stats`
    # reports as the average duration (profiles/), and the *_net fractions are priced with it.  Nothing is subtracted from
    # `achieved` / `frac`.
    for f in fams:
        net = max(f["avg_us"] - (bracket_us or 0.0), 1e-3)
        k = f["avg_us"] / net
        f["avg_us_net"] = net
        f["hbm_frac_net"] = f["hbm_frac"] * k
        f["mfma_frac_net"] = f["mfma_frac"] * k if f["mfma_frac"] is not None else None
    dom = fams[0]
    per_launch_bytes = dom["bytes_per_step"] / dom["launches_per_step"]
    per_launch_flops = dom["flops_per_step"] / dom["launches_per_step"]
    t_hbm = dom["bytes_per_step"] / (HBM_PEAK_GBS * 1e9)
    t_mfma = dom["flops_per_step"] / (dom["mfma_peak_tflops"] * 1e12) if dom["mfma_peak_tflops"] else 0.0
    if t_mfma >= t_hbm:
        roof = {"bound": "mfma", "achieved": dom["tflops"], "peak": dom["mfma_peak_tflops"], "unit": "TFLOP/s",
                "frac": dom["mfma_frac"]}
    else:
        roof = {"bound": "hbm", "achieved": dom["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["hbm_frac"]}
    gemm_t = sum(f["flops_per_step"] / (f["mfma_peak_tflops"] * 1e12) for f in fams if f["mfma_peak_tflops"])
    ideal = sum(f["ideal_fused_bytes_per_step"] for f in fams)
    roof.update(traffic=dom["traffic_bytes_per_launch"], kernel=dom["family"], kernels=dom["kernels"], avg_us=dom["avg_us"],
                ms_per_step=dom["ms_per_step"], launches_per_step=dom["launches_per_step"],
                mfma_frac=dom["mfma_frac"], hbm_frac=dom["hbm_frac"], mfma_peak_tflops=dom["mfma_peak_tflops"],
                event_bracket_us=bracket_us, avg_us_net=dom["avg_us_net"], hbm_frac_net=dom["hbm_frac_net"],
                mfma_frac_net=dom["mfma_frac_net"], algorithmic_bytes_per_launch=per_launch_bytes,
                algorithmic_flops_per_launch=per_launch_flops,
                families=[{k: (round(v, 4) if isinstance(v, float) else v) for k, v in f.items() if k != "kernels"} for f in fams],
                whole_step_lower_bound_ms=1e3 * max(ideal / (HBM_PEAK_GBS * 1e9), gemm_t),
                whole_step_lower_bound_terms_ms={"ideal_fused_bytes_at_hbm_peak": 1e3 * ideal / (HBM_PEAK_GBS * 1e9),
                                                 "gemm_flops_at_matrix_peak": 1e3 * gemm_t})
    return roof


def secondary_runs(steps=30, warmup=8, only=None):
    """Short runs of the other BASELINE configurations (4-scan TFA, AMP, nuScenes shape + AMP) as CHILD processes after
    the headline measurement, so that the driver's default invocation observes them too.  Each entry is the child's
    own JSON line cut down to value / ms_per_step / dtype / config."""
    import subprocess
    out = []
    # (the last three: the reference's own recipes - single-frame at bs 12 / GPU, minkunet_mk34_cr10.yaml:26; FSA with 16 history scans
    # at bs 6 / GPU, minkunet_mk34_cr10_fsa.yaml:14,34, both under --amp as dist_train.sh:18 runs them - and the mask-distillation
    # step, minkunet_mk34_cr10_fsa_kd.yaml; fewer, longer steps)
    runs = [(["--workload", "minkunet_ms"], steps, warmup), (["--amp"], steps, warmup), (["--workload", "nuscenes_ms", "--amp"], steps, warmup),
            (["--eval"], steps, warmup), (["--eval", "--amp"], steps, warmup),
            (["--batch", "12", "--amp"], 15, 4), (["--workload", "minkunet_ms", "--history", "16", "--batch", "6", "--amp"], 12, 4),
            (["--workload", "kd"], 12, 4),
            # TIAF in the reference's mode (UNet2D channels-last under autocast; MIOpen's find-db for its shapes ships with the tree)
            (["--workload", "tiaf", "--amp"], 6, 3)]
    if only:
        runs = [r for r in runs if " ".join(r[0]) in only]
    for extra, steps, warmup in runs:
        note("secondary run: " + " ".join(extra))
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
               "--no-secondary"] + extra           # per-launch events on the first timed step: every entry has its roofline
        entry = {"args": " ".join(extra)}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            rec = None
            for ln in reversed(r.stdout.strip().splitlines()):
                if ln.startswith("{"):
                    rec = json.loads(ln)
                    break
            if r.returncode != 0 or rec is None:
                entry["error"] = (r.stderr or r.stdout)[-400:]
            else:
                entry.update({k: rec.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "loss",
                                                      "host_issue_ms_per_step", "launches_per_step")})
                roof = rec.get("roofline") or {}
                entry["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_us", "avg_us_net", "hbm_frac_net", "mfma_frac_net",
                                                             "ms_per_step", "launches_per_step", "mfma_frac", "hbm_frac",
                                                             "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch",
                                                             "whole_step_lower_bound_ms", "small_layers", "adjoint", "all_maps")} if roof else None
                if rec.get("image_gather") is not None:
                    entry["image_gather"] = rec["image_gather"]
                if rec.get("phases_ms") is not None:
                    entry["phases_ms"] = rec["phases_ms"]
                entry["conv_bytes_per_step"] = rec.get("conv_bytes_per_step")
                entry["ideal_fused_bytes_per_step"] = rec.get("ideal_fused_bytes_per_step")
        except Exception as exc:      # a failed side run must not lose the headline line
            entry["error"] = repr(exc)
        out.append(entry)
    return out


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher: start N ranks as CHILD processes (torch.distributed.run, 127.0.0.1) and
    forward their output.  Runs before this process has made any HIP call (device_count() does not initialise the
    GPU on this image) and never re-execs: the parent only waits."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    share = os.environ.get("TASEG_BENCH_SHARE_DEVICE") == "1"      # rehearsal: ranks share a card (gloo transport only)
    if have < args.gpus and not (share and have >= 1):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} ROCm device(s) visible")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def eval_run(args):
    """`--eval`: K timed evaluation passes of the default workload (MinkUNet mk34 cr 1.0, bs 2, 120 000 points per scan) after W
    warm-up passes; one pass = index plan of the batch + eval-mode forward (running-statistics BatchNorm, no graph) + the eval
    branch's un-voxelisation through inverse_map + arg-max per point, returned as numpy arrays like the reference does
    (minkunet.py:435-455).  One JSON line; value = scans per second."""
    from taseg_amd.data.synthetic import fill_parameters, make_model_cfg
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    torch.cuda.set_device(0)
    batch = args.batch or 2
    points = args.points or 120000
    model = fill_parameters(build_network(make_model_cfg("MinkUNet", in_dim=4, cr=1.0), 20), seed=1).cuda().eval()
    coords, feats, labels, npts = make_scans(0, batch, points, "minkunet")
    offset = torch.tensor([len(coords)], device="cuda", dtype=torch.int32)
    counts = torch.bincount(coords[:, 3].long())
    # one voxel per point after the dataset's sparse_quantize: the inverse map of scene b is the identity on its voxels
    inv = torch.cat([torch.arange(int(c), device="cuda") for c in counts])
    names = [f"scan{b}" for b in range(batch)]

    def make_batch():
        return {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset,
                "targets_mapped": SparseTensor(labels, coords), "inverse_map": SparseTensor(inv, coords), "num_points": counts,
                "name": names}

    # the index plan of batch i + 1 (coordinates only) is staged on a second stream / worker thread while batch i runs - the role
    # of the reference's DataLoader workers; --no-prefetch builds it inline
    from taseg_amd.data.stage import DevicePrefetcher
    # (two batches staged ahead, each on its own stream and thread: one index plan takes longer beside a forward pass than the pass)
    pf = None if args.no_prefetch else DevicePrefetcher(make_batch, model.prepare, threaded=True,
                                                        depth=int(os.environ.get("TASEG_EVAL_STAGE_DEPTH", "2")))

    issue_s = [0.0, 0]

    def issue():
        t_i = time.perf_counter()
        if pf is None:
            bd = make_batch()
        else:
            bd = pf.next()
            pf.prefetch_early()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            out_ = model(bd, defer=True)          # forward + tail enqueued, the arrays collected one pass later
        issue_s[0] += time.perf_counter() - t_i
        issue_s[1] += 1
        return out_

    def passes(n):
        # one batch in flight, as pcseg/eval.py runs the loop: pass i + 1 is issued before the host waits for the arrays of pass i
        pending, out = None, None
        for _ in range(n):
            cur = issue()
            if pending is not None:
                out = pending.result()
            pending = cur
        return pending.result() if pending is not None else out

    out = passes(args.warmup)
    torch.cuda.synchronize()
    host_prof = None
    if os.environ.get("TASEG_BENCH_CPROFILE"):   # diagnostic: interpreter profile of the timed passes
        import cProfile
        host_prof = cProfile.Profile()
        host_prof.enable()
    from taseg_amd import backend as B
    issue_s[0], issue_s[1] = 0.0, 0
    t0 = time.perf_counter()
    out = passes(args.steps)                     # K passes issued AND collected inside the timed region
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # per-launch events around the convolution kernels of a few more passes AFTER the timed region (the roofline of the line)
    roofline = None
    if not args.no_kernel_events:
        B.profile_begin(expected_launches=800)
        passes(3)
        torch.cuda.synchronize()
        records = B.profile_end()
        bracket_us = B.profile_empty_bracket_us() if records else 0.0
        roofline = build_roofline(summarise_profile(records, 3), args.amp, bracket_us)
    if host_prof is not None:
        host_prof.disable()
        host_prof.dump_stats(os.environ["TASEG_BENCH_CPROFILE"])
    assert len(out["point_predict"]) == batch and out["point_predict"][0].shape[0] == int(counts[0])
    ms = 1e3 * dt / args.steps
    print(json.dumps({
        "metric": "scans/sec (eval forward + un-voxelisation + arg-max per point)", "value": batch * args.steps / dt, "unit": "scans/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16-storage/f32-accumulate" if args.amp else "f32", "data": "synthetic", "loss": None,
        "roofline": roofline, "host_issue_ms_per_step": round(1e3 * issue_s[0] / max(issue_s[1], 1), 3), "launches_per_step": launches_of(args),
        "config": {"workload": f"MinkUNet mk34 cr1.0 evaluation pass (eval-mode BatchNorm, no graph), bs={batch}, voxel 0.05 m, "
                               f"{'autocast fp16' if args.amp else 'fp32'}, index plan + forward + un-voxelisation + arg-max",
                   "points_per_step_per_gpu": int(npts), "voxels_per_step_per_gpu": int(coords.shape[0])}}), flush=True)


def main():
    args = parse()
    if getattr(args, "eval", False):
        return eval_run(args)
    if args.cpu_leg:                       # child of cpu_baseline(): no GPU, one timed CPU pass, one JSON record
        from taseg_amd.data.synthetic import make_model_cfg
        ms = args.workload in ("minkunet_ms", "nuscenes_ms")
        cfg = make_model_cfg("MinkUNetMs" if ms else "MinkUNet", in_dim=5 if ms else 4, cr=1.0)
        print(json.dumps(_cpu_leg(cfg, args.points or 120000, args.cpu_sector_deg, args.cpu_leg, args.cpu_passes,
                                  args.cpu_warmup)), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or let "
                         f"`bench.py --gpus N` start the ranks itself)")
    dog = Watchdog(rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the product path has no CPU fallback)")
    local %= max(torch.cuda.device_count(), 1)            # R/train.py:247-251: device_ids=[LOCAL_RANK % ngpu]
    torch.cuda.set_device(local)
    backend = os.environ.get("TASEG_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm; "gloo" only to rehearse N ranks on one card
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            from taseg_amd.options import options as _opts
            _opts.syncbn_single_rank = True      # the collective path on a one-rank group
        dog.beat("init_process_group")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    dog.beat("build model and inputs")

    from taseg_amd import backend as B
    from taseg_amd import planes as _planes
    from taseg_amd.data.synthetic import make_model_cfg
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor

    B.set_conv_impl(args.conv_impl)
    if args.workload == "tiaf":
        args.no_kernel_events = True      # (its roofline entry is the image gather's, measured after the timed steps)
    nusc = args.workload == "nuscenes_ms"
    tiaf = args.workload == "tiaf"
    kd = args.workload == "kd"
    ms = args.workload in ("minkunet_ms", "nuscenes_ms", "tiaf", "kd")
    # (kd: the per-launch events cover BOTH networks of the step - the frozen teacher's forward pass and the student's forward + backward)
    if args.batch is None:
        args.batch = 4 if nusc else 6 if kd else 2       # (kd yaml :44 BATCH_SIZE_PER_GPU 6)
    if args.points is None:
        args.points = 34700 if nusc else 120000
    voxel = 0.1 if nusc else VOXEL
    num_class = 17 if nusc else 20
    name = "MinkUNetMsMm" if tiaf else "MinkUNetMsKd" if kd else "MinkUNetMs" if ms else "MinkUNet"
    # nuScenes FSA feeds 4 features (the time flag column is cut by IN_FEATURE_DIM: 4, nuscenes fsa yaml:16,27)
    extra_cfg = {}
    if tiaf:
        from taseg_amd.data.synthetic import TIAF_CFG
        from taseg_amd.options import options as _opts
        extra_cfg = dict(TIAF_CFG)
        miopen_find_db()
        if os.environ.get("TASEG_BENCH_MIOPEN_FIND") == "1":       # MIOpen's measured solver search for UNet2D's convolutions (minutes
            torch.backends.cudnn.benchmark = True                  # on a machine without a find-db for these shapes)
        if args.image_layout:
            _opts.image_layout = args.image_layout
    if kd:
        extra_cfg = dict(SAMPLING_TYPE="random", MAX_VOXEL=3000, FEAT_KD="mse", FEAT_KD_WEIGHT=10.0)      # kd yaml :35-38
    cfg = make_model_cfg(name, in_dim=(4 if nusc else 5) if ms else 4, cr=1.0, if_dist=use_dist and not args.local_bn, **extra_cfg)
    torch.manual_seed(0)
    model = build_network(cfg, num_class).cuda().train()
    if kd:
        model.fix_part_param()           # the teacher's parameters are frozen (R/train.py calls it for this model; minkunet_ms_kd.py:719-722)
    net = model
    reducer = None
    lr, mom, wd = 0.02 * args.batch * world, 0.9, 1e-4
    flat = not args.torch_optim
    if flat:
        # flat-bucket SGD (taseg_amd.optim): gradients land in flat buckets during backward (all-reduced over ranks on
        # their own communicator when N > 1), unscale + clip + SGD + loss-scale update in 3 launch kinds, no host read
        from taseg_amd.optim import FlatSGD
        # the buckets always get a communicator of their own (launched in bucket-index order on every rank); SyncBatchNorm's
        # statistics go through the default group (TASEG_DIST_SINGLE_COMM=1, default) or the library-owned communicator
        group = dist.new_group(backend=backend) if use_dist else None
        opt = FlatSGD(model, lr=lr, momentum=mom, weight_decay=wd, max_norm=10.0, amp=args.amp, process_group=group)
    else:
        if use_dist and not args.torch_ddp:
            from taseg_amd.parallel import GradBucketReducer
            reducer = GradBucketReducer(model, process_group=dist.new_group(backend=backend))
        elif use_dist:
            # buffers (BN running statistics) are identical on every rank by construction (SyncBatchNorm): no broadcast
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], gradient_as_bucket_view=True,
                                                            broadcast_buffers=False, static_graph=True)
        opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=mom, weight_decay=wd)

    nvox = [0]
    if ms:
        from taseg_amd.data.stage import build_multiscan_batch
        from taseg_amd.data.synthetic import FLEXIBLE_STEPS_KITTI, FLEXIBLE_STEPS_NUSC
        if nusc:
            from taseg_amd.data.nuscenes import build_nuscenes_batch
            nsamples, npts, n_sweeps = make_nusc_samples(rank, args.batch, args.points)

            def make_batch():
                bd = build_nuscenes_batch(nsamples, voxel, FLEXIBLE_STEPS_NUSC)
                nvox[0] = int(bd["lidar_ms"].C.shape[0])
                return bd
        elif tiaf:
            from taseg_amd.data.tiaf import build_tiaf_batch, build_tiaf_sample
            tiaf_frames, tiaf_proj, npts = make_tiaf_frames(rank, args.batch, args.points)
            n_fov = [0]

            def make_batch():
                samples = [build_tiaf_sample(fr, FLEXIBLE_STEPS_KITTI, TIAF_MULTISCAN, TIAF_STEP_IMAGE, tiaf_proj,
                                             (TIAF_HEIGHT, TIAF_WIDTH), voxel, name=f"{rank}/{b}") for b, fr in enumerate(tiaf_frames)]
                bd = build_tiaf_batch(samples)
                nvox[0] = int(bd["lidar_ms"].C.shape[0])
                n_fov[0] = int(bd["lidar_fov_ms"].C.shape[0])
                return bd
        elif kd:
            # MULTISCAN 16, ONLY_HISTORY (kd yaml :17-21): the student's cloud keeps the history points whose PSEUDO label's class is
            # due at that frame offset, the teacher's those whose ANNOTATION is (semantickitti_ms_kd.py:140-147, 290-358)
            scans, npts = make_multiscans(rank, args.batch, args.points, history=16, pseudo_flip=0.1)
            scans_gt = [{k: v for k, v in s.items() if k != "pseudo"} for s in scans]
            nvox_gt = [0]

            def make_batch():
                bd = build_multiscan_batch(scans, voxel, FLEXIBLE_STEPS_KITTI)
                bd["lidar_ms_gt"] = build_multiscan_batch(scans_gt, voxel, FLEXIBLE_STEPS_KITTI)["lidar_ms"]
                nvox[0] = int(bd["lidar_ms"].C.shape[0])
                nvox_gt[0] = int(bd["lidar_ms_gt"].C.shape[0])
                return bd
        else:
            scans, npts = make_multiscans(rank, args.batch, args.points, history=args.history or 4)

            def make_batch():
                bd = build_multiscan_batch(scans, voxel, FLEXIBLE_STEPS_KITTI)
                nvox[0] = int(bd["lidar_ms"].C.shape[0])
                return bd
    else:
        coords, feats, labels, npts = make_scans(rank, args.batch, args.points, args.workload)
        offset = torch.tensor([len(coords)], device="cuda", dtype=torch.int32)
        nvox[0] = int(coords.shape[0])

        _cached = {}

        def make_batch():
            bd = {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset}
            if os.environ.get("TASEG_REUSE_PLAN") == "1":      # diagnostic: what the step costs without any staging work
                if "_plan" in _cached:
                    bd["_plan"] = _cached["_plan"]
                else:
                    _cached["_plan"] = model.prepare(bd)
            return bd

    from taseg_amd.data.stage import DevicePrefetcher
    prepare = (lambda bd: bd.get("_plan") or model.prepare(bd)) if os.environ.get("TASEG_REUSE_PLAN") == "1" else model.prepare
    threaded = os.environ.get("TASEG_STAGE_THREAD", "1" if args.amp else "0") == "1"
    pf = None if args.no_prefetch else DevicePrefetcher(make_batch, prepare, threaded=threaded,
                                                        depth=int(os.environ.get("TASEG_STAGE_DEPTH", "1")))
    # host-bound steps (the threaded stage is their mark) start staging batch i + 1 right after the forward pass of step i
    early_stage = os.environ.get("TASEG_STAGE_EARLY", "1" if threaded else "0") == "1"

    scaler = torch.amp.GradScaler("cuda", enabled=args.amp)
    B.planes_in_use = _planes._ENABLED and not args.amp        # names the 128-column fp32 pair GEMM in the kernel table

    opt_events = []

    # TASEG_BENCH_HOST_PHASES=1 (diagnostic): host time the training thread spends ISSUING each phase of a step (no device
    # synchronisation: what a host-bound line is made of), printed to stderr after the run
    # host time the training thread spends ISSUING each phase of a step (no device synchronisation: what a host-bound line is made
    # of; seven clock reads per step).  Always collected: `host_issue_ms_per_step` of the JSON line; TASEG_BENCH_HOST_PHASES=1
    # prints the phases to stderr after the run
    host_phases = {}

    def _hp(name, t):
        now = time.perf_counter()
        host_phases[name] = host_phases.get(name, 0.0) + now - t
        return now

    def step(time_optimizer=False):
        # one step = stage one batch (rulebooks / index plan; for minkunet_ms also the temporal aggregation and
        # voxelisation) + forward + loss + backward + clip + SGD.  With the prefetcher the batch staged inside
        # step i is the one step i+1 trains on (every timed step still stages exactly one batch).
        th = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            if pf is None:
                ret, _, _ = net(make_batch())
            else:
                bd_ = pf.next()
                th = _hp("next_batch", th)
                ret, _, _ = net(bd_)
        th = _hp("forward", th)
        if pf is not None and early_stage:
            pf.prefetch_early()          # the next batch is staged beside this step's backward pass (worker thread)
        loss = ret["loss"].float().mean()
        th = _hp("loss", th)
        if flat:
            (loss * opt.loss_scale()).backward()
        else:
            scaler.scale(loss).backward()
        th = _hp("backward", th)
        if time_optimizer:               # BASELINE's metric wants the optimizer's share stated: events on sampled steps
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if flat:
            opt.step()                   # reducer.finish() + grad stats + decide + apply, all on the device
        else:
            if reducer is not None:
                reducer.finish()
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
            scaler.step(opt)
            scaler.update()
        if time_optimizer:
            ev[1].record()
            opt_events.append(ev)
        th = _hp("optimizer", th)
        if pf is not None:
            pf.prefetch()
        th = _hp("stage_next", th)
        host_phases["steps"] = host_phases.get("steps", 0) + 1
        return ret["loss"]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # weight gradients on a second stream (csrc/block.hip, taseg_amd/_fast.py): both settings are timed on this model, batch and
    # machine BEFORE the warm-up (16 steps) and the faster one is kept - like a convolution-algorithm search, outside the contract's
    # W + K steps.  With several ranks every rank times its own (the second stream is rank-local; the gradient buckets join it
    # before their all-reduce, parallel.GradBucketReducer._launch).
    wgrad_side = None
    if flat and not args.no_wgrad_tune:
        from taseg_amd import _fast
        wgrad_side = _fast.tune_wgrad_stream(step, fence)
        if rank == 0:
            note(f"weight gradients on a second stream: {wgrad_side[0]} (step {wgrad_side[1]} ms without, {wgrad_side[2]} ms with)")
    if rank == 0:
        note(f"{name} {args.workload}: {args.warmup} warm-up + {args.steps} timed steps on {world} rank(s)")
    for i in range(args.warmup):
        dog.beat(f"warm-up step {i}")
        step()
    dog.beat("fence after warm-up")
    fence()
    if dist is not None:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # every rank: push RCCL's C-stdio version banner out before the result line
    if not args.no_kernel_events:
        B.profile_begin(expected_launches=800 * len(range(0, args.steps, EVENT_EVERY)))   # ~330 (bs 2) .. per step
    host_phases.clear()
    from taseg_amd import _fast as _f
    if _f.module() is not None:
        _f.module().host_times()
    host_prof = None
    if os.environ.get("TASEG_BENCH_CPROFILE") and rank == 0:      # diagnostic: interpreter profile of the timed steps
        import cProfile
        host_prof = cProfile.Profile()
        host_prof.enable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not args.no_kernel_events:
            B.profile_pause(i % EVENT_EVERY != 0)      # HIP events around the conv kernels of every EVENT_EVERY-th timed step
        dog.beat(f"timed step {i}")           # (one attribute store: nothing the timed region notices)
        profiled = not args.no_kernel_events and i % EVENT_EVERY == 0
        # the per-launch figures of the roofline are durations of kernels that have the device to themselves: the profiled step
        # (one in EVENT_EVERY) keeps its weight gradients on the caller's stream; on the second stream they run BESIDE the input
        # gradient and both read longer
        if profiled and wgrad_side is not None and wgrad_side[0]:
            _fast.wgrad_stream(False)
        loss = step(time_optimizer=profiled)
        if profiled and wgrad_side is not None and wgrad_side[0]:
            _fast.wgrad_stream(True)
    dog.beat("fence after the timed steps")
    fence()
    dt = time.perf_counter() - t0
    if host_prof is not None:
        host_prof.disable()
        host_prof.dump_stats(os.environ["TASEG_BENCH_CPROFILE"])
    n_hp = max(host_phases.pop("steps", 1), 1)
    host_issue = {k: 1e3 * v / n_hp for k, v in host_phases.items()}
    if os.environ.get("TASEG_BENCH_HOST_PHASES") == "1" and rank == 0:
        note("host issue time per step (ms, timed steps): " + ", ".join(f"{k} {v:.2f}" for k, v in host_issue.items()))
        if _f.module() is not None:      # the native block / stage nodes' own clocks (reset after the warm-up)
            nf, tf, af, nb, tb, ab = _f.module().host_times()
            note(f"native nodes per step: forward {nf / n_hp:.0f} blocks {1e-6 * tf / n_hp:.2f} ms ({1e-6 * af / n_hp:.2f} in the backend "
                 f"calls), backward {nb / n_hp:.0f} blocks {1e-6 * tb / n_hp:.2f} ms ({1e-6 * ab / n_hp:.2f} in the backend calls)")
    dog.beat("collect")
    records = B.profile_end()
    # what an event pair measures with nothing between its two records, after the timed region: the per-launch figures
    # below are NOT corrected by it (in the flow of a step the bracket costs less: rocprofv3 puts the dominant kernel
    # ~3 us below its event figure, the idle-stream bracket reads ~4.7 us) - it is reported as context for that gap
    bracket_us = B.profile_empty_bracket_us() if records else 0.0
    profiled_steps = len(range(0, args.steps, EVENT_EVERY))
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = args.batch * world * args.steps / dt
    bus = None
    if dist is not None and world > 1:
        # the achieved bus bandwidth of the step's one exchange, the gradient all-reduce, measured
        # on its own after the timed region (a flat fp32 buffer of the model's size, bus = 2 (n - 1) / n * bytes / t)
        n_par = sum(p.numel() for p in model.parameters())
        buf = torch.zeros(n_par, dtype=torch.float32, device="cuda")
        for _ in range(2):
            dist.all_reduce(buf)
        fence()
        t1 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t1) / 5], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bus = {"bytes": 4 * n_par, "ms": 1e3 * float(t.item()),
               "bus_GBps": 2 * (world - 1) / world * 4 * n_par / float(t.item()) / 1e9}
        del buf

    dog.stop()                                # what follows is rank 0's reporting (CPU baseline legs, side runs: minutes, own limits)
    if rank == 0:
        prof = summarise_profile(records, profiled_steps)
        roofline = build_roofline(prof, args.amp, bracket_us)
        if roofline is not None:
            roofline["small_layers"] = small_layer_share(records, profiled_steps)
        if roofline is not None and wgrad_side is not None:
            roofline["weight_gradients_on_second_stream"] = bool(wgrad_side[0])
            roofline["profiled_steps"] = ("weight gradients on the caller's stream (kernels timed one at a time); the other timed steps "
                                          "run them on the second stream, beside the input gradient") if wgrad_side[0] else "as timed"
        line = {
            "metric": "scans/sec (train fwd+bwd) at ~120k pts/scan" if not nusc else
            "scans/sec (train fwd+bwd), nuScenes-shaped sweeps", "value": value, "unit": "scans/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 storage / f32 accumulate (torch.autocast)" if args.amp else "f32", "data": "synthetic",
            "config": {"workload": f"{name} mk34 cr1.0 ("
                                   f"{'nuScenes FSA stage: sweeps selected at 1 m of driven distance up to 15 m + keyframes' if nusc else 'TIAF: 16 fused history scans + 5 camera frames of 384 x 1280 per sample, UNet2D + UNet3D + fusion head, five losses' if tiaf else 'mask distillation: 16 history scans fused twice per sample, frozen teacher forward + student step, segmentation + feature MSE losses' if kd else f'{args.history or 4}-scan TFA multi-scan' if ms else 'single-frame'}), "
                                   f"bs={args.batch}/GPU, voxel {voxel:g} m, {'AMP fp16' if args.amp else 'fp32'}, rulebook+fwd+loss+bwd+SGD step",
                       "points_per_step_per_gpu": npts, "voxels_per_step_per_gpu": nvox[0],
                       "parallelism": f"dp{world}",
                       # chosen by a timing of both settings before the warm-up (None: not applicable / not tuned)
                       "wgrad_on_second_stream": None if wgrad_side is None else bool(wgrad_side[0])},
            "loss": float(loss.detach()),
            # inside ms_per_step; gradient clipping + SGD (world > 1: + the tail of the bucketed all-reduce it waits for)
            # host time per step the training thread spends issuing the step (forward + loss + backward + optimizer: everything but
            # waiting for the staged batch and staging the next one in line) - against ms_per_step: how far the line is from host-bound
            "host_issue_ms_per_step": round(sum(v for k, v in host_issue.items() if k in ("forward", "loss", "backward", "optimizer")), 3),
            "host_issue_phases_ms": {k: round(v, 3) for k, v in host_issue.items()},
            "launches_per_step": launches_of(args),
            "grad_allreduce": bus,
            "optimizer_ms_per_step": (sum(a.elapsed_time(b) for a, b in opt_events) / len(opt_events)) if opt_events else None,
            "roofline": roofline,
            "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in prof[:8]],
            # the convolution kernels of one step: algorithmic bytes of the design as built (two passes: per-pair Z round
            # trip included) next to SURVEY.md section 8(d)'s ideal-fused lower bound of the same layers
            "conv_bytes_per_step": sum(r["bytes_per_step"] for r in prof) if prof else None,
            "ideal_fused_bytes_per_step": sum(r["ideal_fused_bytes_per_step"] for r in prof) if prof else None,
        }
        if tiaf:
            from taseg_amd.options import options as _opts
            line["config"]["fov_points_per_step_per_gpu"] = n_fov[0]
            line["config"]["image_layout"] = _opts.image_layout
            # (a FRESH batch: the forward pass of a step overwrites the FOV cloud's feature matrix, pixel columns included)
            gat = tiaf_gather_roofline(model, make_batch(), _opts.image_layout, torch.float16 if args.amp else torch.float32)
            line["image_gather"] = gat
            # the line's roofline object: the hand-written kernel of this workload that moves the most bytes - the gather of the
            # 96-channel full-resolution map - forward and adjoint (HBM-bound: algorithmic bytes / event-bracketed launch time)
            u4 = gat[0]
            line["roofline"] = {"bound": "hbm", "kernel": "image_rows_gather_kernel / image_rows_scatter_kernel (u4)" if _opts.image_layout == "nhwc"
                                else "image_gather_rows_kernel / image_scatter_rows_kernel (u4)",
                                "achieved": u4["forward"]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": u4["forward"]["hbm_frac"],
                                "traffic": None, "avg_us": u4["forward"]["avg_us"], "algorithmic_bytes_per_launch": u4["forward"]["algorithmic_bytes"],
                                "adjoint": u4["adjoint"], "all_maps": [{"map": g_["map"], "forward_hbm_frac": round(g_["forward"]["hbm_frac"], 3),
                                                                         "adjoint_hbm_frac": round(g_["adjoint"]["hbm_frac"], 3)} for g_ in gat]}
            if flat:
                line["phases_ms"] = tiaf_phase_table(model, net, opt, make_batch, args.amp)
        if kd:
            line["config"]["teacher_voxels_per_step_per_gpu"] = nvox_gt[0]
        if world == 1 and not args.no_cpu_baseline and not nusc and not tiaf and not kd:      # cpu_baseline is defined on the KITTI-shaped scan
            line["cpu_baseline"] = cpu_baseline(args, name, cfg.IN_FEATURE_DIM)
            if line["cpu_baseline"]["value"]:
                line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        plain = (world == 1 and not args.force_dist and args.workload == "minkunet" and not args.amp and args.conv_impl == 0
                 and not args.torch_optim and args.batch == 2 and args.points == 120000)
        if plain and not args.no_secondary:
            line["secondary"] = secondary_runs()
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL prints its version banner through C stdio: flush it BEFORE the line
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        from taseg_amd import rccl
        rccl.shutdown()                  # library-owned SyncBatchNorm communicators
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
