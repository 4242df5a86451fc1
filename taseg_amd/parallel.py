"""Data-parallel helpers: one process per GPU, `torch.distributed` over RCCL (backend "nccl" on ROCm).

The path shards by whole scans (batches are block-diagonal in the batch index):
every rank trains on its own scans and the only data-path exchange per step is the gradient
all-reduce (DDP buckets overlapped with backward) plus SyncBN statistics when IF_DIST is set.
The reference's per-step logging collectives (commu_utils.average_reduce_value x3, six blocking
all-gathers) are deliberately not reproduced.
"""
import os
from typing import List

import torch
import torch.distributed as dist

__all__ = ["env_rank", "init_distributed", "shard_seeds", "wrap_ddp", "reduce_max", "reduce_confusion"]


def env_rank():
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_distributed(backend: str = "nccl"):
    """Rendezvous from the torchrun environment (reference: common_utils.init_dist_pytorch)."""
    rank, local, world = env_rank()
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return rank, local, world


def shard_seeds(rank: int, world: int, batch: int, epoch: int = 0) -> List[int]:
    """Scan seeds of this rank for one step: seed = 1000 * sequence + frame with sequence = rank, so ranks
    never share a scan (the role DistributedSampler plays in the reference, data/__init__.py:134-141)."""
    assert 0 <= rank < world
    return [1000 * rank + 10 * (epoch * batch + b) for b in range(batch)]


def wrap_ddp(model: torch.nn.Module, local_rank=None):
    """DistributedDataParallel with bucket views (gradients are reduced in place, overlapped with backward)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    ids = None if local_rank is None else [local_rank]
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, gradient_as_bucket_view=True)


def reduce_max(value: float, device="cpu") -> float:
    """max over ranks (bench timing)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_confusion(hist: torch.Tensor) -> torch.Tensor:
    """Sum per-rank confusion matrices with ONE all-reduce (the reference pickles them through the shared
    file system and two barriers, common_utils.merge_results_dist)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist
