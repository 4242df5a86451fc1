"""Library-owned RCCL communicators for SyncBatchNorm (csrc/rccl.hip).

`direct_comm(group)` returns an opaque communicator handle over the ranks of a torch.distributed process group, or
None when the direct path is unavailable (TASEG_RCCL_DIRECT=0, librccl.so not loadable, a failed self-test on any
rank) - callers then fall back to `dist.all_reduce` on the group.  Creation is collective: every rank of the group
reaches it at the same point (the first SyncBatchNorm forward of the first step).  The 128-byte RCCL id travels over
the process group itself; before the communicator is trusted one all-reduce is checked against the group's own, and
the ranks agree on the outcome, so either all of them use the direct path or none does.
"""
import ctypes
import os
import warnings

import torch
import torch.distributed as dist

from . import _lib as L

_comms = {}


def _agree(ok, group, dev):
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def _create(group):
    lib = L.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    ok = True
    try:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L.check(lib.ts_rccl_load(path.encode() if os.path.exists(path) else None), "ts_rccl_load")
    except Exception as e:  # noqa: BLE001 - any failure means "use the process group instead"
        warnings.warn(f"taseg_amd: direct RCCL path unavailable ({e}); SyncBatchNorm uses torch.distributed")
        ok = False
    if not _agree(ok, group, dev):
        return None
    rank, nranks = dist.get_rank(group), dist.get_world_size(group)
    idbuf = (ctypes.c_ubyte * 128)()
    if rank == 0:
        L.check(lib.ts_rccl_unique_id(idbuf), "ts_rccl_unique_id")
    idt = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=dev)
    dist.broadcast(idt, src=dist.get_global_rank(group, 0), group=group)
    idbytes = (ctypes.c_ubyte * 128)(*idt.cpu().tolist())
    comm = ctypes.c_void_p()
    rc = lib.ts_rccl_comm_init(idbytes, nranks, rank, ctypes.byref(comm))
    ok = rc == 0 and bool(comm.value)
    if ok:
        # self-test against the process group's own all-reduce (sum of rank-dependent doubles)
        t = torch.tensor([rank + 1.0, 1.0, 0.5 * rank], dtype=torch.float64, device=dev)
        want = t.clone()
        rc = lib.ts_rccl_allreduce_f64(comm, L.ptr(t), 3, L.stream())
        dist.all_reduce(want, group=group)
        ok = rc == 0 and torch.equal(t, want)
    if not _agree(ok, group, dev):
        if comm.value:
            lib.ts_rccl_comm_destroy(comm)
        warnings.warn("taseg_amd: direct RCCL communicator failed its self-test; SyncBatchNorm uses torch.distributed")
        return None
    return comm


def single_communicator() -> bool:
    """TASEG_DIST_SINGLE_COMM (default 1): SyncBatchNorm's statistics all-reduces go through torch.distributed's default
    process group (c10d, on the compute stream) like nn.SyncBatchNorm under DDP in the reference (R/train.py:247-251); the
    gradient buckets always use a communicator of their own (parallel.GradBucketReducer), launched in bucket-index order.
    By default the library issues them on that group's own communicator itself (`_borrow`: the same RCCL calls on the same
    communicator and stream, without c10d's per-call dispatch).  TASEG_DIST_SINGLE_COMM=0 moves the statistics onto a communicator
    CREATED by this module - not the default until it has run with more than one rank on real devices
    (tests/test_gpu_dist.py::test_two_ranks_rccl needs two GPUs and has been skipped on every box so far)."""
    return os.environ.get("TASEG_DIST_SINGLE_COMM", "1") != "0"


def _borrow(group):
    """The process group's OWN RCCL communicator (ProcessGroupNCCL._comm_ptr) as the handle of the direct path: the statistics
    all-reduces are then the same `ncclAllReduce` calls on the same communicator and the same (current) stream that
    `c10d_sum` makes through the dispatcher - issued from the library instead (126 calls per training step, ~10-16 us of host time
    each through c10d).  Nothing is created or bootstrapped here; the handle is never destroyed by this module.  None where the
    group's backend has no such communicator (gloo) or the self-test fails on any rank."""
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = L.load()
    backend = None
    try:
        backend = group._get_backend(dev) if hasattr(group, "_get_backend") else None
    except Exception:  # noqa: BLE001 - a group without a device backend
        backend = None
    if backend is None or not hasattr(backend, "_comm_ptr"):
        return None                                       # (a property of the group's type: the same answer on every rank)

    def step(fn):
        """one rank-local step; afterwards the ranks agree whether ALL of them succeeded (every rank issues the same collectives
        in the same order whatever happens locally)"""
        ok = True
        try:
            ok = bool(fn())
        except Exception as e:  # noqa: BLE001 - any failure means "go through torch.distributed"
            warnings.warn(f"taseg_amd: the process group's communicator cannot be used directly ({e}); SyncBatchNorm goes through "
                          f"torch.distributed")
            ok = False
        return _agree(ok, group, dev)

    def load():
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L.check(lib.ts_rccl_load(path.encode() if os.path.exists(path) else None), "ts_rccl_load")
        return True

    if not step(load):
        return None
    # the communicator of a device exists after the group's first collective on it
    t = torch.tensor([dist.get_rank(group) + 1.0, 1.0], dtype=torch.float64, device=dev)
    want = t.clone()
    c10d_sum(want, group)
    comm = ctypes.c_void_p()

    def handle():
        comm.value = int(backend._comm_ptr())
        return bool(comm.value)

    if not step(handle):
        return None
    # one all-reduce through the library on the borrowed handle against the group's own result
    if not step(lambda: lib.ts_rccl_allreduce_f64(comm, L.ptr(t), 2, L.stream()) == 0 and torch.equal(t, want)):
        return None
    _borrowed.add(id(group))
    return comm


_borrowed = set()


def direct_comm(group):
    """Communicator handle (ctypes.c_void_p) for `group`, resolved on first use (collectively: every rank reaches the first
    SyncBatchNorm forward); None = go through torch.distributed (`c10d_sum`).
      default (TASEG_DIST_SINGLE_COMM=1, TASEG_RCCL_DIRECT unset): the group's OWN communicator, called from the library (`_borrow`) -
        the calls c10d would make, without its dispatch; where that is not available (gloo), torch.distributed;
      TASEG_RCCL_DIRECT=0: always through torch.distributed;
      TASEG_RCCL_DIRECT=1 or TASEG_DIST_SINGLE_COMM=0: a communicator created by this module (`_create`)."""
    key = id(group)
    if key not in _comms:
        want = os.environ.get("TASEG_RCCL_DIRECT")
        if want == "0":
            _comms[key] = None
        elif want == "1" or not single_communicator():
            _comms[key] = _create(group)
        else:
            _comms[key] = _borrow(group)
    return _comms[key]


def c10d_sum(buf, group):
    """In-place sum of `buf` over the ranks of `group` on the group's own communicator; the current stream waits for the
    collective (no host wait).  ProcessGroup.allreduce directly: torch.distributed.all_reduce adds ~10 us of argument
    checking per call, and a SyncBatchNorm training step makes 126 of these.  asyncOp = False as in dist.all_reduce(...,
    async_op=False): ProcessGroupNCCL then issues the collective on the CURRENT stream - no side stream, no pair of
    cross-stream hand-overs per call."""
    global _sync_opts
    if _sync_opts is None:
        _sync_opts = dist.AllreduceOptions()
        _sync_opts.asyncOp = False
    work = group.allreduce([buf], _sync_opts)
    if work is not None:             # (ProcessGroupNCCL returns no work object for a current-stream collective)
        work.wait()


_sync_opts = None


def shutdown():
    """Destroy the communicators and drop the native node's process-group handles (call before
    dist.destroy_process_group())."""
    from . import _fast
    from .torchsparse.nn import modules as _modules
    if _fast._mod is not None:
        _fast._mod.clear_groups()
    _modules._group_ids.clear()
    lib = L.load()
    for key, comm in list(_comms.items()):
        if comm is not None and comm.value and key not in _borrowed:      # (a borrowed communicator belongs to its process group)
            lib.ts_rccl_comm_destroy(comm)
        del _comms[key]
    _borrowed.clear()
