"""The N > 1 path on CPU: world_size 2 (and 3, ragged) over gloo - shard, gather, reduce."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from trajopt_amd.sharding import gather_problem_vector, reduce_progress, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(N, world, rank)
        p = torch.arange(lo, hi, dtype=torch.float64)
        cost = 100.0 + p * 0.5                      # what this rank "solved"
        status = (torch.arange(lo, hi) % 4).to(torch.int32)
        lam = 1.0 + p
        full_cost = gather_problem_vector(cost, N)
        full_status = gather_problem_vector(status, N)
        prog = reduce_progress(cost, status, lam)
        q.put((rank, full_cost.numpy(), full_status.numpy(), prog))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 64), (2, 7), (3, 10)])
def test_gather_and_reduce_over_gloo(world, N):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx = np.arange(N)
    for rank, cost, status, prog in got:
        np.testing.assert_array_equal(cost, 100.0 + idx * 0.5)         # every rank holds the full result
        np.testing.assert_array_equal(status, idx % 4)
        assert prog["total_cost"] == pytest.approx((100.0 + idx * 0.5).sum())
        assert prog["nb_stopped"] == int((idx % 4 != 0).sum()) and prog["nb_gave_up"] == int((idx % 4 == 3).sum())
        assert prog["max_lmbda"] == float(N)


def test_single_process_path_needs_no_process_group():
    import torch
    from trajopt_amd.sharding import gather_problem_vector, reduce_progress
    c = torch.tensor([1.0, 2.0, float("-inf")], dtype=torch.float64)
    assert torch.equal(gather_problem_vector(c, 3), c)
    pr = reduce_progress(c, torch.tensor([0, 2, 3], dtype=torch.int32), torch.tensor([1.0, 5.0, 2.0], dtype=torch.float64))
    assert pr == dict(total_cost=3.0, nb_stopped=2, nb_gave_up=1, max_lmbda=5.0)
