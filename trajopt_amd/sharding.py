"""Multi-GPU sharding of independent problems: one process per GPU, rank r owns the contiguous
block [r*N/G, (r+1)*N/G) of the problem axis, no exchange during an iteration.  The reference does the same
fan-out with one OS process per problem (joblib/loky, /root/reference/examples/ilqr/mpc/pendulum_parallel.py:40-45).

The only collectives are the ones the path really has: a gather of per-problem results (cost, status,
iteration counts: KB-scale) and an all-reduce of four counters for a global progress report.  They run on
torch.distributed - backend "nccl" (= RCCL over xGMI) on GPU tensors, "gloo" on CPU tensors in the tests.
"""
import numpy as np


def shard_bounds(nb_problems, world_size, rank):
    """[lo, hi) of rank's contiguous block; the first nb_problems % world_size ranks get one extra problem."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(int(nb_problems), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(array, world_size, rank, axis=0):
    lo, hi = shard_bounds(array.shape[axis], world_size, rank)
    idx = [slice(None)] * array.ndim
    idx[axis] = slice(lo, hi)
    return array[tuple(idx)]


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def _comm_tensor(t, group=None):
    """`t` on the device the process group's backend communicates from: RCCL ("nccl") takes GPU tensors as they
    are, gloo takes host tensors (several ranks may share one GPU in a rehearsal: their results cross on the CPU)."""
    dist = _dist()
    if dist is not None and t.is_cuda and dist.get_backend(group) != "nccl":
        return t.cpu()
    return t


def gather_problem_vector(local, nb_problems, group=None):
    """All ranks receive the full [nb_problems] vector assembled from the per-rank blocks (ragged blocks are
    padded to the largest one for the all_gather).  `local` is a 1-D torch tensor; the result lives on the
    backend's device (the GPU under RCCL, the host under gloo)."""
    import torch
    dist = _dist()
    if dist is None:
        assert local.numel() == nb_problems
        return local.clone()
    local = _comm_tensor(local, group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(nb_problems, world, rank)
    assert local.numel() == hi - lo, "rank %d holds %d problems, expected %d" % (rank, local.numel(), hi - lo)
    width = -(-nb_problems // world)
    buf = torch.zeros(width, dtype=local.dtype, device=local.device)
    buf[:hi - lo] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = torch.empty(nb_problems, dtype=local.dtype, device=local.device)
    for r in range(world):
        a, b = shard_bounds(nb_problems, world, r)
        out[a:b] = parts[r][:b - a]
    return out


def reduce_progress(last_return, status, lmbda, group=None):
    """Global [sum of costs, #stopped (status != 0), #gave up (status == 3), max lambda] over all ranks."""
    import torch
    dist = _dist()
    last_return, status, lmbda = (_comm_tensor(t, group) for t in (last_return, status, lmbda))
    fin = torch.where(torch.isfinite(last_return), last_return, torch.zeros_like(last_return))
    sums = torch.stack([fin.sum(), (status != 0).sum().to(fin.dtype), (status == 3).sum().to(fin.dtype)])
    mx = lmbda.max().reshape(1).to(fin.dtype) if lmbda.numel() else torch.zeros(1, dtype=fin.dtype, device=fin.device)
    if dist is not None:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return dict(total_cost=float(sums[0]), nb_stopped=int(sums[1]), nb_gave_up=int(sums[2]), max_lmbda=float(mx[0]))


class ShardedILQR:
    """N problems split over the ranks of the default process group; each rank runs its block on its own GPU - as
    one BatchedILQR or, where that is faster (larger models: trajopt_amd/ilqr/streamed.py; `sub_batches` None = its rule,
    1 = never), as a few sub-batches on HIP streams of their own.  `init_state` (N, dm_state) and `kff_init` (N, dm_act, T)
    are the GLOBAL arrays (every rank passes the same ones and slices its block), so results do not depend on the number
    of GPUs."""

    def __init__(self, env, nb_steps, init_state, kff_init=None, sub_batches=None, **kw):
        from .ilqr import StreamedILQR
        dist = _dist()
        self.world = dist.get_world_size() if dist else 1
        self.rank = dist.get_rank() if dist else 0
        init_state = np.asarray(init_state, dtype=np.float64)
        self.nb_problems = init_state.shape[0]
        self.lo, self.hi = shard_bounds(self.nb_problems, self.world, self.rank)
        if kff_init is None:
            rng = np.random.Generator(np.random.PCG64(1337))
            kff_init = 1e-4 * rng.standard_normal((self.nb_problems, env.unwrapped.dm_act, nb_steps))
        self.local = StreamedILQR(env, nb_steps, init_state[self.lo:self.hi],
                                  kff_init=np.asarray(kff_init)[self.lo:self.hi], sub_batches=sub_batches, **kw)

    def run(self, nb_iter=25):
        """Runs the local block, then gathers -> dict(last_return, status, iters) of all N problems + progress."""
        self.local.run(nb_iter=nb_iter)
        b = self.local
        last_return, status = b.vector("last_return"), b.vector("status")
        out = dict(last_return=gather_problem_vector(last_return, self.nb_problems),
                   status=gather_problem_vector(status, self.nb_problems),
                   iters=gather_problem_vector(b.vector("iters"), self.nb_problems))
        out["progress"] = reduce_progress(last_return, status, b.vector("lmbda"))
        return out
