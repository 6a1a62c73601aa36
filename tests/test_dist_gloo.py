"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: per-sample sharding and the gradient all-reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sola_amd import dist as sdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _lr, w = sdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    mine = sdist.shard_indices(11, rank, world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 70000, 3, 1)]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    n_coll = sdist.allreduce_gradients(params, world, bucket_bytes=100000)
    expect = [(1 + 2) / 2 * (i + 1) for i in range(4)]
    ok = all(torch.allclose(p.grad, torch.full_like(p, e)) for p, e in zip(params, expect))
    gathered = sdist.gather_scores({i: i * i for i in mine}, rank, world)
    dist.barrier()
    q.put((rank, mine, n_coll, ok, gathered))
    dist.destroy_process_group()


def test_two_rank_shard_and_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    (r0, m0, c0, ok0, g0), (r1, m1, c1, ok1, g1) = res
    assert sorted(m0 + m1) == list(range(11)) and not set(m0) & set(m1)  # every sample owned exactly once
    assert m0 == [0, 2, 4, 6, 8, 10]
    assert ok0 and ok1 and c0 == c1 == 3  # identical bucket schedule on both ranks: [5], [70000], [3,1]
    merged = {}
    for d in g0:
        merged.update(d)
    assert merged == {i: i * i for i in range(11)} and g0 == g1


def test_single_rank_is_noop():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    assert sdist.allreduce_gradients([p], world=1) == 0
    assert sdist.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
