"""SyncBatchNorm's statistics all-reduces: which transport carries them (csrc/rccl.hip holds the direct ones).

DEFAULT: torch.distributed - `c10d_sum` on the caller's process group, the collective the reference's nn.SyncBatchNorm issues
(R/train.py:247-251, R/pcseg/model/segmentor/voxel/minkunet/minkunet.py:23-25).  It is the only transport that has c10d's
per-communicator mutex, work tracking and watchdog around it, and the one every multi-process test of this tree runs.

`direct_comm(group)` returns an opaque communicator handle for the two OPT-IN transports (None = the default), chosen with
`options.rccl_direct` (environment: TASEG_RCCL_DIRECT):
  "borrow" - the process group's OWN RCCL communicator (`ProcessGroupNCCL._comm_ptr()`, a private accessor: guarded by a torch
             version check), the `ncclAllReduce` issued by this library on the current stream: 126 c10d dispatches per step saved
             (~1 ms of host time on the one-rank line);
  "create" - a communicator this module creates over the ranks of the group (the 128-byte RCCL id travels over the group).
Both are collective to set up (every rank reaches the first SyncBatchNorm forward), self-test one all-reduce against the
group's own and agree on the outcome over the group, so either all ranks use the direct path or none does.  Neither has run
with more than one rank on real devices (tests/test_gpu_dist.py::test_two_ranks_rccl needs two GPUs): they stay opt-in until
one has.
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
        _load(lib, path if os.path.exists(path) else None)
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


# torch releases whose ProcessGroupNCCL._comm_ptr() has been checked to return the ncclComm_t of the current device
_BORROW_TORCH = ("2.9", "2.10")


def _torch_rccl_path():
    """librccl.so of the running PyTorch if it is ALREADY mapped into this process (RTLD_NOLOAD), else None: a handle borrowed
    from torch's RCCL must only ever be passed to that same RCCL image"""
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        return None
    try:
        ctypes.CDLL(path, mode=os.RTLD_NOW | os.RTLD_NOLOAD)
    except OSError:
        return None
    return path


_loaded_path = [None]


def _load(lib, path):
    """bind the library's RCCL entry points to `path` (None: the default search path); refuses a second, different image"""
    if _loaded_path[0] is not None and _loaded_path[0] != (path or ""):
        raise RuntimeError(f"RCCL already bound to '{_loaded_path[0]}', not '{path or ''}'")
    L.check(lib.ts_rccl_load(path.encode() if path else None), "ts_rccl_load")
    _loaded_path[0] = path or ""


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
    if not torch.__version__.startswith(tuple(v + "." for v in _BORROW_TORCH)):
        warnings.warn(f"taseg_amd: ProcessGroupNCCL._comm_ptr() is unchecked on torch {torch.__version__}; SyncBatchNorm goes "
                      f"through torch.distributed")
        return None                                       # (the same torch on every rank)

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
        # only the RCCL image torch itself runs on: the handle is meaningless to any other
        path = _torch_rccl_path()
        if path is None:
            raise RuntimeError("torch's own librccl.so is not mapped into this process")
        _load(lib, path)
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
    SyncBatchNorm forward); None = go through torch.distributed (`c10d_sum`), the default.  See the module docstring."""
    key = id(group)
    if key not in _comms:
        from .options import options
        want = options.rccl_direct
        if want == "create":
            _comms[key] = _create(group)
        elif want == "borrow":
            _comms[key] = _borrow(group)
        else:
            _comms[key] = None
        _groups[key] = group          # (keeps the group object alive, so that its id is not reused while the entry exists)
    return _comms[key]


_groups = {}


def forget(group):
    """drop the entry of a process group that is being destroyed (a created communicator is destroyed, a borrowed one is the
    group's own)"""
    key = id(group)
    comm = _comms.pop(key, None)
    _groups.pop(key, None)
    if comm is not None and comm.value and key not in _borrowed:
        L.load().ts_rccl_comm_destroy(comm)
    _borrowed.discard(key)


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
    _groups.clear()
    _borrowed.clear()
