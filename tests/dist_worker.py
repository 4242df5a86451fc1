"""One rank of the N > 1 parity check (a script, started by tests/test_gpu_dist.py; not collected by pytest).

    RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT   rendezvous, as torch.distributed.run sets them
    TASEG_DIST_BACKEND   "nccl" (RCCL, one GPU per rank) or "gloo" (ranks may share one GPU: device = LOCAL_RANK % count)
    OUT                  directory for rank<r>.npz

Every rank builds the same MinkUNet (SyncBatchNorm, as all reference configs: IF_DIST True), trains one step on ITS OWN
scan (seed 41 + rank) through the data-parallel path of bench.py - FlatSGD's GradBucketReducer all-reduce overlapped
with backward, SyncBatchNorm statistics over the ranks - and stores logits, the averaged gradients and the BatchNorm
running statistics.  The test compares them with ONE process run on the concatenated batch (R/train.py:247-251: DDP
averages the per-rank gradients of the per-rank mean losses).

With backend nccl and > 1 rank it also (a) repeats the step with the SyncBatchNorm collectives on the process group
instead of the library-owned RCCL communicator and stores both, (b) times the gradient all-reduce of an mk34-sized
buffer (151.5 MB) and prints the bus bandwidth.
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VOXEL = 0.05


def make_scan(seed, batch_index=0, n_points=20000):
    from taseg_amd.data.synthetic import synth_scan
    from taseg_amd.torchsparse.utils.quantize import sparse_quantize
    pts, lab = synth_scan(seed, n_points=n_points, n_beams=32, n_az=1000)
    pc = np.round(pts[:, :3] / VOXEL).astype(np.int32)
    pc -= pc.min(0)
    _, idx, _ = sparse_quantize(pc, return_index=True, return_inverse=True)
    coords = np.concatenate([pc[idx], np.full((len(idx), 1), batch_index, np.int32)], 1)
    return coords, pts[idx], lab[idx].astype(np.int64)


def build(if_dist, seed=3):
    from taseg_amd.data.synthetic import fill_parameters, make_model_cfg
    from taseg_amd.pcseg.model import build_network
    cfg = make_model_cfg("MinkUNet", in_dim=4, cr=0.5, if_dist=if_dist)       # mk34 depth, half width
    return fill_parameters(build_network(cfg, 20), seed=seed).cuda().train()


def scan_loss(logits, labels):
    # per-scan mean cross-entropy (ignore 0): the quantity every rank back-propagates; DDP averages over ranks
    return torch.nn.functional.cross_entropy(logits, labels, ignore_index=0)


def one_step(model, batches, process_group=None, amp=False):
    """forward + backward over `batches` = [(coords, feats, labels)] (one per scan of this process); returns logits,
    gradients (after the bucketed all-reduce), BatchNorm running statistics"""
    from taseg_amd.optim import FlatSGD
    from taseg_amd.torchsparse import SparseTensor
    opt = FlatSGD(model, lr=0.0, momentum=0.0, weight_decay=0.0, process_group=process_group)
    if os.environ.get("TASEG_WORKER_SIDE") == "1":      # asked for: weight gradients on the second stream.  Under N > 1 the request is
        from taseg_amd import _fast                      # refused (taseg_amd._fast.require_single_stream), on one rank it is granted
        granted = _fast.wgrad_stream(True)
        assert granted == (opt.reducer.world == 1), (granted, opt.reducer.world)
    coords = torch.from_numpy(np.concatenate([b[0] for b in batches])).cuda()
    feats = torch.from_numpy(np.concatenate([b[1] for b in batches])).cuda()
    labels = torch.from_numpy(np.concatenate([b[2] for b in batches])).cuda()
    grabbed = {}
    h = model.classifier.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o))
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):      # the reference's default mode (dist_train.sh --amp)
        model({"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": torch.tensor([0])})
    h.remove()
    logits = grabbed["logits"].float()
    sizes = [len(b[0]) for b in batches]
    parts, lparts = torch.split(logits, sizes), torch.split(labels, sizes)
    loss = sum(scan_loss(a, b) for a, b in zip(parts, lparts)) / len(batches)
    opt.zero_grad()
    scale = 1024.0 if amp else 1.0          # static loss scale: half-storage gradients of the early layers would underflow
    (loss * scale).backward()
    opt.reducer.finish()
    grads = {n: (p.grad.detach().float() / scale).cpu().numpy() for n, p in model.named_parameters()}
    stats = {n: b.detach().float().cpu().numpy() for n, b in model.named_buffers() if "running" in n}
    return logits.detach().float().cpu().numpy(), grads, stats, float(loss)


def pack(prefix, logits, grads, stats, loss):
    out = {prefix + "logits": logits, prefix + "loss": np.float64(loss)}
    out.update({prefix + "grad/" + k: v for k, v in grads.items()})
    out.update({prefix + "stat/" + k: v for k, v in stats.items()})
    return out


def unused_parameter_steps(rank, group):
    """TASEG_WORKER_MODE=unused: FlatSGD over ranks where a parameter gets its gradient on rank 0 ONLY and another one on
    no rank: three optimizer steps (momentum, weight decay), the parameters after each."""
    from taseg_amd.optim import FlatSGD
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 5)).cuda()
    net.register_parameter("one_rank_only", torch.nn.Parameter(torch.ones(5, device="cuda")))
    net.register_parameter("nobody", torch.nn.Parameter(torch.ones(3, device="cuda")))
    opt = FlatSGD(net, lr=0.1, momentum=0.9, weight_decay=1e-2, max_norm=10.0, process_group=group)
    out = {}
    for step in range(3):
        opt.zero_grad()
        g = torch.Generator().manual_seed(7 * step + rank)
        x, y = torch.randn(6, 8, generator=g).cuda(), torch.randint(0, 5, (6,), generator=g).cuda()
        logits = net[2](net[1](net[0](x)))
        if rank == 0:
            logits = logits + net.one_rank_only
        torch.nn.functional.cross_entropy(logits, y).backward()
        opt.step()
        for n, p in net.named_parameters():
            out[f"step{step}/{n}"] = p.detach().cpu().numpy().copy()
    return out


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("TASEG_DIST_BACKEND", "nccl")
    dev = int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if os.environ.get("TASEG_WORKER_MODE") == "unused":
        np.savez(os.path.join(os.environ["OUT"], f"rank{rank}.npz"), **unused_parameter_steps(rank, dist.new_group(backend=backend)))
        dist.barrier()
        dist.destroy_process_group()
        return
    from taseg_amd import rccl
    out = {}
    scan = make_scan(41 + rank)
    amp = os.environ.get("TASEG_WORKER_AMP") == "1"
    group = dist.new_group(backend=backend)      # the buckets' own communicator, as bench.py passes it
    # TASEG_WORKER_LOCAL_BN=1: plain BatchNorm (per-rank statistics, bench.py --local-bn) instead of the configs' SyncBatchNorm
    out.update(pack("", *one_step(build(os.environ.get("TASEG_WORKER_LOCAL_BN") != "1"), [scan], group, amp=amp)))
    direct = rccl.direct_comm(dist.group.WORLD)
    out["direct_rccl"] = np.int64(1 if direct is not None else 0)
    out["borrowed"] = np.int64(1 if id(dist.group.WORLD) in rccl._borrowed else 0)
    if backend == "nccl" and world > 1:
        # (a) the same step with the SyncBatchNorm collectives through torch.distributed
        rccl._comms[id(dist.group.WORLD)] = None
        out.update(pack("c10d/", *one_step(build(True), [scan], group)))
        rccl._comms[id(dist.group.WORLD)] = direct
        # (b) gradient all-reduce of an mk34-sized flat buffer (37.88 M fp32): bus bandwidth = 2 (n-1)/n bytes / t
        buf = torch.ones(37_882_900, device="cuda")
        for _ in range(3):
            dist.all_reduce(buf, group=group)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        iters = 10
        for _ in range(iters):
            dist.all_reduce(buf, group=group)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        bus = 2 * (world - 1) / world * buf.numel() * 4 / dt / 1e9
        if rank == 0:
            print(json.dumps({"allreduce_bytes": buf.numel() * 4, "ranks": world, "ms": 1e3 * dt, "bus_GBps": bus}), flush=True)
        out["allreduce_bus_GBps"] = np.float64(bus)
    np.savez(os.path.join(os.environ["OUT"], f"rank{rank}.npz"), **out)
    dist.barrier()
    rccl.shutdown()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
