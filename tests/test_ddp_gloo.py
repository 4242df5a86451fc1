"""N > 1 path on CPU: two gloo processes exercise the data-parallel helpers (scan sharding, DDP gradient
averaging == single process on the concatenated batch, max-over-ranks timing, confusion all-reduce)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from taseg_amd import parallel as P


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 5))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = P.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    seeds = P.shard_seeds(rank, world, batch=2)
    net = P.wrap_ddp(_net())
    g = torch.Generator().manual_seed(seeds[0])
    x, y = torch.randn(6, 8, generator=g), torch.randint(0, 5, (6,), generator=g)
    torch.nn.functional.cross_entropy(net(x), y).backward()
    grads = [p.grad.clone() for p in net.parameters()]
    tmax = P.reduce_max(float(rank + 1))
    hist = P.reduce_confusion(torch.full((3, 3), rank + 1, dtype=torch.int64))
    out.put((rank, seeds, [g.tolist() for g in grads], tmax, hist.tolist()))   # plain lists: no shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get() for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, s0, g0, t0, h0), (_, s1, g1, t1, h1) = res
    assert not set(s0) & set(s1)                                   # ranks never share a scan
    g0, g1 = [torch.tensor(a) for a in g0], [torch.tensor(b) for b in g1]
    h0, h1 = torch.tensor(h0), torch.tensor(h1)
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b)                                # DDP left identical averaged grads on both ranks
    # ... equal to the single-process gradient of the mean of the two per-rank losses
    net = _net()
    total = 0
    for seeds in (s0, s1):
        g = torch.Generator().manual_seed(seeds[0])
        x, y = torch.randn(6, 8, generator=g), torch.randint(0, 5, (6,), generator=g)
        total = total + torch.nn.functional.cross_entropy(net(x), y) / 2
    total.backward()
    for a, p in zip(g0, net.parameters()):
        assert torch.allclose(a, p.grad, atol=1e-6)
    assert t0 == t1 == 2.0
    assert int(h0[0, 0]) == 3 and torch.equal(h0, h1)


def test_single_process_helpers_are_noops():
    assert P.shard_seeds(3, 8, 2) == [3000, 3010]
    m = _net()
    assert P.wrap_ddp(m) is m and P.reduce_max(1.5) == 1.5


def _reducer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    P.init_distributed(backend="gloo")
    net = _net()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)                         # rank 1 starts elsewhere: the reducer must broadcast rank 0's weights
    extra = torch.nn.Parameter(torch.ones(3))       # a parameter that never gets a gradient
    net.register_parameter("unused", extra)
    # tiny buckets: several collectives per step; a dedicated group, as bench.py / FlatSGD pass it
    red = P.GradBucketReducer(net, process_group=dist.new_group(backend="gloo"), bucket_mb=0.0002)
    w0 = [p.detach().clone() for p in net.parameters()]
    rows = []
    for step in range(2):                           # two steps: hooks must re-arm
        for p in net.parameters():
            p.grad = None
        g = torch.Generator().manual_seed(100 * step + rank)
        x, y = torch.randn(6, 8, generator=g), torch.randint(0, 5, (6,), generator=g)
        torch.nn.functional.cross_entropy(net[2](net[1](net[0](x))), y).backward()
        red.finish()
        rows.append([p.grad.clone().tolist() for p in net.parameters()])
    out.put((rank, [w.tolist() for w in w0], rows))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_reducer_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get() for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w_a, rows_a), (_, w_b, rows_b) = res
    assert w_a == w_b                                              # broadcast: both ranks hold rank 0's weights
    for step in range(2):
        net = _net()
        net.register_parameter("unused", torch.nn.Parameter(torch.ones(3)))
        total = 0.0
        for rank in range(world):
            g = torch.Generator().manual_seed(100 * step + rank)
            x, y = torch.randn(6, 8, generator=g), torch.randint(0, 5, (6,), generator=g)
            total = total + torch.nn.functional.cross_entropy(net[2](net[1](net[0](x))), y) / world
        total.backward()
        want = [p.grad if p.grad is not None else torch.zeros_like(p) for p in net.parameters()]
        for got_a, got_b, w in zip(rows_a[step], rows_b[step], want):
            assert torch.allclose(torch.tensor(got_a), w, atol=1e-6) and torch.allclose(torch.tensor(got_b), w, atol=1e-6)


def _flags_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    P.init_distributed(backend="gloo")
    net = _net()
    net.register_parameter("one_rank_only", torch.nn.Parameter(torch.ones(5)))     # rank 0 alone puts it into its loss
    net.register_parameter("nobody", torch.nn.Parameter(torch.ones(3)))            # no rank ever uses it
    red = P.GradBucketReducer(net, process_group=dist.new_group(backend="gloo"), bucket_mb=0.0002)
    names = [n for n, _ in net.named_parameters()]
    rows = []
    for step in range(2):
        for p in net.parameters():
            p.grad = None
        g = torch.Generator().manual_seed(7 * step + rank)
        x, y = torch.randn(6, 8, generator=g), torch.randint(0, 5, (6,), generator=g)
        logits = net[2](net[1](net[0](x)))
        if rank == 0:
            logits = logits + net.one_rank_only
        torch.nn.functional.cross_entropy(logits, y).backward()
        red.finish()
        flags, local_unused = {}, []
        for b in red.buckets:
            for i, p in enumerate(b["params"]):
                name = names[[id(q) for q in net.parameters()].index(id(p))]
                flags[name] = float(b["flags"][i])
                if i in b["unused"]:
                    local_unused.append(name)
        rows.append((flags, sorted(local_unused), net.one_rank_only.grad.tolist(), net.nobody.grad.tolist()))
    out.put((rank, rows))
    dist.barrier()
    dist.destroy_process_group()


def test_unused_parameter_decision_is_global():
    """A parameter without a gradient on ONE rank only still takes the averaged gradient everywhere (DDP's
    find_unused_parameters rule); the all-reduced flag that says so is identical on every rank, step after step."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_flags_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get() for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, rows0), (_, rows1) = res
    for (f0, u0, g0, n0), (f1, u1, g1, n1) in zip(rows0, rows1):
        assert f0 == f1                                            # the same decision on both ranks
        assert f0["one_rank_only"] == 0.5 and f0["nobody"] == 0.0 and f0["0.weight"] == 1.0
        assert u0 == ["nobody"] and u1 == ["nobody", "one_rank_only"]      # the rank-local lists differ ...
        assert g0 == g1 and any(abs(v) > 0 for v in g0)                    # ... the averaged gradient does not
        assert n0 == n1 == [0.0, 0.0, 0.0]


def _order_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    P.init_distributed(backend="gloo")
    net = _net()
    # registered LAST = first bucket (reverse registration order): rank 0 alone uses it, so on rank 0 the bucket is complete
    # early in backward while rank 1 only completes it in finish()
    tail = torch.nn.Module()
    tail.register_parameter("one_rank_only", torch.nn.Parameter(torch.ones(5)))
    net.add_module("tail", tail)                    # (a module's own parameters come FIRST in parameters(), a child's last)
    # ONE parameter per bucket and NO process group given: the reducer makes its own; buckets of equal padded size exist
    # (two biases of 16-element slots), so a rank-local launch order would pair different buckets across ranks
    red = P.GradBucketReducer(net, bucket_mb=1e-7)
    assert red.group is not None and len(red.buckets) == len(list(net.parameters()))
    order, launch = [], red._launch
    ids = [id(b) for b in red.buckets]
    red._launch = lambda b: (order.append(ids.index(id(b))), launch(b))[1]
    rows = []
    for step in range(2):
        for p in net.parameters():
            p.grad = None
        g = torch.Generator().manual_seed(11 * step + rank)
        x, y = torch.randn(6, 8, generator=g), torch.randint(0, 5, (6,), generator=g)
        logits = net[2](net[1](net[0](x)))
        if rank == 0:
            logits = logits + tail.one_rank_only
        torch.nn.functional.cross_entropy(logits, y).backward()
        during = list(order)
        red.finish()
        rows.append((during, list(order), [p.grad.clone().tolist() for p in net.parameters()]))
        order.clear()
    out.put((rank, rows))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_launch_order_is_the_same_on_every_rank():
    """One parameter per bucket, one parameter used by rank 0 only (round-3 advice): the all-reduces must still pair up -
    every rank launches its buckets in index order, whatever the order in which its gradients became complete."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get() for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, rows0), (_, rows1) = res
    for (during0, all0, g0), (during1, all1, g1) in zip(rows0, rows1):
        assert all0 == all1 == sorted(all0)                 # index order on both ranks
        assert len(during0) > len(during1) == 0             # rank 1 could launch nothing before finish(): bucket 0 was incomplete
        assert g0 == g1                                     # the same averaged gradients
        assert any(abs(v) > 0 for v in g0[-1])              # one_rank_only (last parameter, bucket 0) took rank 0's gradient / 2
