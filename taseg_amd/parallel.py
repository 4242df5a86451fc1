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
from .options import options

__all__ = ["env_rank", "init_distributed", "shard_seeds", "wrap_ddp", "reduce_max", "reduce_confusion",
           "GradBucketReducer"]


_grad_epoch = 0


def grad_epoch() -> int:
    """Counter of "gradient slots may be handed out again" events (GradBucketReducer.finish(), FlatSGD.zero_grad()): a producer that
    writes a gradient straight into a bucket slot (the stage programs, csrc/fastpath/stage_program.h) claims a parameter's slot at
    most once per epoch - the second use of a weight in one graph gets a fresh tensor and autograd adds the two."""
    return _grad_epoch


def bump_grad_epoch():
    global _grad_epoch
    _grad_epoch += 1


# parameter -> the reducer that owns its bucket slot.  Kept OUTSIDE the tensors: a weak reference in a Parameter's __dict__ would make
# torch.save(model) / pickle fail (Parameter.__reduce_ex__ pickles __dict__).  id-keyed, checked against the live object on look-up.
_reducer_of = {}


def reducer_of(param):
    """the live GradBucketReducer whose bucket holds `param`'s gradient slot, or None"""
    ref = _reducer_of.get(id(param))
    red = ref() if ref is not None else None
    if red is None:
        return None
    slot = red._slot_of.get(id(param))
    return red if slot is not None and red.buckets[slot[0]]["params"][slot[1]] is param else None


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


class GradBucketReducer:
    """Gradient averaging over ranks in a few flat buckets, overlapped with backward - what DDP's reducer does,
    minus its per-parameter device copies.  With `gradient_as_bucket_view` DDP copies every freshly produced
    gradient into its bucket separately (380 parameters -> 380 small kernels, 1.6 ms per step here); this reducer
    waits until the last gradient of a bucket exists, moves the whole bucket with ONE multi-tensor copy, re-points
    `p.grad` at the bucket views (no copy back), and all-reduces the flat buffer asynchronously.

        reducer = GradBucketReducer(model)            # after the process group is up; broadcasts rank 0's weights
        loss.backward()                               # hooks launch a bucket's all-reduce when it is complete
        reducer.finish()                              # before clip / optimizer.step(): wait for the collectives

    The buckets' all-reduces run on a communicator of their OWN: `process_group`, or - when none is given and there is
    more than one rank - a group created here (every rank constructs its reducer, so every rank makes the collective
    `dist.new_group()` call).  SyncBatchNorm's statistics all-reduces run during backward on the default group / the
    library-owned communicator; sharing one communicator would make the program order of bucket and statistics
    collectives depend on when each rank's gradients happen to be complete.
    Launch order is the bucket index on every rank: bucket i is launched only after buckets 0..i-1 (a bucket that is
    complete earlier waits; what backward leaves incomplete - parameters unused on this rank - is launched by finish()
    in index order).  Which parameters receive a gradient may differ between ranks; the sequence of collectives may not.
    Buckets follow reverse registration order (roughly the order gradients appear).  Parameters that received no
    gradient in a step are reduced as zeros (every rank must contribute the same buffer) and listed in
    bucket["unused"] so that the optimizer can leave them untouched like torch.optim.SGD does.  Contract: ONE backward
    pass per finish(); a second one raises.  Works with any backend (RCCL on GPU, gloo in the CPU tests)."""

    def __init__(self, model: torch.nn.Module, process_group=None, bucket_mb: float = 32.0, broadcast: bool = True):
        # also usable without a process group (one process): the buckets then only flatten the gradients
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if live else 1
        if live and process_group is None and self.world > 1 and not options.dist_buckets_on_default_group:
            process_group = dist.new_group(backend=dist.get_backend())
        self.group = process_group
        if self.world > 1:          # (see _fast.require_single_stream: no weight gradients on a second stream under N > 1)
            from . import _fast
            _fast.require_single_stream("gradient buckets are averaged over %d ranks" % self.world)
        self._next = 0              # index of the next bucket to launch (launch order = index order on every rank)
        self._slot_of = {}          # id(parameter) -> (bucket index, position in the bucket)
        params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_mb * (1 << 20):
                self._add_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._add_bucket(cur)
        self._works = []
        if broadcast and self.world > 1:
            for b in self.buckets:                  # every rank starts from rank 0's parameters (DDP does the same)
                torch._foreach_copy_(b["views"], [p.data for p in b["params"]])
                dist.broadcast(b["flat"], src=dist.get_global_rank(self.group, 0) if self.group else 0,
                               group=self.group)
                torch._foreach_copy_([p.data for p in b["params"]], b["views"])
        for b in self.buckets:
            for p in b["params"]:
                p.register_post_accumulate_grad_hook(self._make_hook(b))

    ALIGN = 16      # elements: every tensor starts on a 64-byte boundary of the flat buffer (kernels that take
                    # parameters, e.g. the BatchNorm ones, require 16-byte aligned pointers; padding stays zero)

    def _add_bucket(self, params):
        first = params[0]
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        # [gradients (off elements) | one "this rank has a gradient" flag per parameter]: one buffer, one all-reduce
        n_flags = -(-len(params) // self.ALIGN) * self.ALIGN
        flat_all = torch.zeros(off + n_flags, dtype=first.dtype, device=first.device)
        flat, flags = flat_all[:off], flat_all[off:off + len(params)]
        views = [flat[o:o + p.numel()].view_as(p) for o, p in zip(offsets, params)]
        import weakref
        bucket_index = len(self.buckets)
        for i, (p, v) in enumerate(zip(params, views)):
            # producers that can write a gradient wherever they are told (the conv block nodes: their weight gradient is
            # a full overwrite) put it straight into the bucket; _launch then finds p.grad already in place and copies
            # nothing for it (the convolution weights are 99 % of the gradient bytes)
            p._taseg_grad_dest = v
            # ... and producers that deliver a whole stage's gradients themselves (deliver below) find their way back here
            # through reducer_of(p)
            _reducer_of[id(p)] = weakref.ref(self)
            self._slot_of[id(p)] = (bucket_index, i)
        self.buckets.append({"params": list(params), "flat": flat, "flat_all": flat_all, "flags": flags, "views": views,
                             "offsets": offsets, "pending": len(params), "launched": False, "unused": []})

    def _make_hook(self, bucket):
        def hook(_param):
            if bucket["launched"] or bucket["pending"] <= 0:
                # a second backward before finish() (gradient accumulation, retain_graph): the bucket's all-reduce is
                # already in flight and p.grad aliases the buffer it reduces - adding into it now would race with
                # the collective and leave the ranks with different gradients
                raise RuntimeError("GradBucketReducer: a gradient arrived for a bucket whose all-reduce was already "
                                   "launched - exactly one backward pass per finish() / optimizer step is supported")
            bucket["pending"] -= 1
            # in index order only: a complete bucket behind an incomplete one waits for it (or for finish())
            while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
                self._launch(self.buckets[self._next])
                self._next += 1
        return hook

    def check_open(self, params):
        """raises unless every bucket of `params` still waits for gradients (called by a direct producer BEFORE it writes into the
        slots: a bucket whose all-reduce is in flight must not be written to)"""
        for p in params:
            bucket = self.buckets[self._slot_of[id(p)][0]]
            if bucket["launched"] or bucket["pending"] <= 0:
                raise RuntimeError("GradBucketReducer: a backward pass reached a bucket whose all-reduce was already launched - "
                                   "exactly one backward pass per finish() / optimizer step is supported")

    def deliver(self, params):
        """Gradients of `params` have been written straight into their bucket slots by a producer that bypasses autograd's
        AccumulateGrad for them (the stage programs, csrc/fastpath/stage_program.h: one call per stage and backward pass instead of
        one hook per parameter): p.grad becomes the slot's view and the buckets count down exactly as the hooks would have."""
        for p in params:
            b, i = self._slot_of[id(p)]
            bucket = self.buckets[b]
            if bucket["launched"] or bucket["pending"] <= 0:
                raise RuntimeError("GradBucketReducer: a gradient arrived for a bucket whose all-reduce was already "
                                   "launched - exactly one backward pass per finish() / optimizer step is supported")
            p.grad = bucket["views"][i]
            bucket["pending"] -= 1
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, bucket):
        # weight gradients on a second stream (taseg_amd/_fast.py) are complete THERE: this stream - on which the bucket is scaled and
        # whose work the all-reduce waits for - first waits for everything handed to the second stream so far (one event)
        from . import _fast
        _fast.join_wgrad_stream()
        src, dst = [], []
        bucket["unused"] = [i for i, p in enumerate(bucket["params"]) if p.grad is None]
        for p, v in zip(bucket["params"], bucket["views"]):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(bucket["params"], bucket["views"]):
            p.grad = v                               # the optimizer reads the reduced values in place
        if self.world > 1:
            bucket["flags"].fill_(1.0)               # (the previous step's all-reduce left averages here)
            for i in bucket["unused"]:
                bucket["flags"][i] = 0.0
            bucket["flat_all"].div_(self.world)
            self._works.append(dist.all_reduce(bucket["flat_all"], group=self.group, async_op=True))
        bucket["launched"] = True

    def finish(self):
        """Launch what backward left incomplete (unused parameters), wait for every collective, re-arm."""
        for b in self.buckets[self._next:]:          # index order, like the launches during backward
            if not b["launched"]:
                self._launch(b)
        self._next = 0
        bump_grad_epoch()
        for w in self._works:
            w.wait()
        self._works = []
        for b in self.buckets:
            b["pending"], b["launched"] = len(b["params"]), False
            for p in b["params"]:
                p._taseg_dest_claimed = False        # the slots may be handed out again (nn/modules.py::_claim_grad_dest)
