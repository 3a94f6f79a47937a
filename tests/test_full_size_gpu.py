"""The hot path at BASELINE.json's FULL sizes (C2-C5), checked through size-independent properties:

* problems are independent, so a batch made of `reps` copies of D distinct problems must return `reps` bitwise
  identical copies of D results, whatever the grid, wave and workgroup a copy lands in (this is what exercises the
  full-size launch: offsets beyond 2^31 bytes, every workgroup, the XCD mapping);
* the D distinct results are compared with the CPU oracle at the full horizon, to the 1e-9 bar.

Inputs are tiled on the device; only the first copy travels back.
"""
import numpy as np
import pytest

from tests.util import TOL, assert_close

pytestmark = pytest.mark.gpu

KEYS = ("Cxx", "cx", "Cuu", "cu", "Cxu", "A", "B")


def _tile(t, reps, layout):
    return t.repeat(*((reps,) + (1,) * (t.dim() - 1))) if layout == 0 else t.repeat(*((1,) * (t.dim() - 1) + (reps,)))


def _copies_identical(t, D, reps, layout):
    """All `reps` copies of the D-problem block equal the first one, bit for bit (NaN-free outputs)."""
    import torch
    if layout == 0:
        v = t.reshape((reps, D) + tuple(t.shape[1:]))
        return bool(torch.equal(v, v[:1].expand_as(v)))
    v = t.reshape(tuple(t.shape[:-1]) + (reps, D))
    return bool(torch.equal(v, v[..., :1, :].expand_as(v)))


def _first(t, D, layout):
    from trajopt_amd import layout as L
    return L.from_device((t[:D] if layout == 0 else t[..., :D]).contiguous(), layout)


@pytest.mark.parametrize("cfg,n,m,T,N,D,layout,full", [
    ("C2", 4, 1, 500, 4096, 64, 1, False),      # four-problems-per-wave matrix-core kernel
    ("C2", 4, 1, 500, 4096, 64, 1, True),
    ("C3", 6, 1, 500, 8192, 64, 1, False),      # lane-per-problem kernel, one GPU's share of C3
    ("C4", 12, 4, 500, 4096, 8, 0, False),      # wave-per-problem matrix-core kernel
])
def test_ilqr_backward_full_size(oracle, cfg, n, m, T, N, D, layout, full):
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    reps = N // D
    stack = S.backward_inputs(n, m, T, D, seed=1337)
    d = {k: _tile(L.to_device(stack[k], layout), reps, layout) for k in KEYS}
    lam = torch.ones(N, dtype=torch.float64, device="cuda")
    o = backward.ilqr_backward_pass(*[d[k] for k in KEYS], lam, 1, n, m, T, N, layout, full=full)
    torch.cuda.synchronize()
    assert int((o["diverge"] != -1).sum()) == 0
    names = ("K", "kff", "dV") + (("Qxx", "Qux", "Quu", "qx", "qu", "V", "v") if full else ())
    for k in names:
        assert _copies_identical(o[k], D, reps, layout), "%s %s: copies differ" % (cfg, k)
    got = {k: _first(o[k], D, layout) for k in names}
    idx = {"Qxx": 0, "Qux": 1, "Quu": 2, "qx": 3, "qu": 4, "V": 5, "v": 6, "dV": 7, "K": 8, "kff": 9}
    for p in range(0, D, max(1, D // 8)):
        q = S.problem(stack, p)
        ref, failed = oracle.backward_pass(*[q[k] for k in KEYS], 1.0, 1, n, m, T)
        assert not failed
        for k in names:
            assert_close(got[k][p], ref[idx[k]], "%s %s[p=%d]" % (cfg, k, p))


def test_riccati_sweep_full_size_c5(oracle):
    """C5: 1024 x (64 + 16), T = 1000, PROBLEM_MAJOR - two distinct problems, 512 copies (about 130 GB of HBM)."""
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    n, m, T, N, D, layout = 64, 16, 1000, 1024, 2, 0
    reps = N // D
    stack = S.backward_inputs(n, m, T, D, seed=1337, affine=True)
    keys = KEYS + ("c",)
    d = {k: _tile(L.to_device(stack[k], layout), reps, layout) for k in keys}
    o = backward.riccati_backward_pass(*[d[k] for k in keys], n, m, T, N, layout)
    torch.cuda.synchronize()
    del d
    assert int(o["status"].abs().sum()) == 0
    for k in ("K", "kff", "V", "v"):
        assert _copies_identical(o[k], D, reps, layout), "C5 %s: copies differ" % k
    got = {k: _first(o[k], D, layout) for k in ("K", "kff", "V", "v")}
    del o
    torch.cuda.empty_cache()
    for p in range(D):
        q = S.problem(stack, p)
        ref = oracle.riccati_backward_pass(*[q[k] for k in keys], n, m, T)
        for k, r in zip(("K", "kff", "V", "v"), ref):
            assert_close(got[k][p], r, "C5 %s[p=%d]" % (k, p))


def test_ilqr_iteration_full_size_c2(oracle):
    """The bench workload itself - 4096 cartpoles, T = 500, 11 step sizes - as 64 copies of 64 distinct problems:
    two full iterations; every copy bitwise the first, the first against the oracle's driver."""
    from oracle import oracle as O
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    N, D, T, iters = 4096, 64, 500, 2
    reps = N // D
    pe, oe = envs.make("cartpole", dt=0.02, uw=1e-5), O.make_env("cartpole", dt=0.02, uw=[1e-5])
    rng = np.random.Generator(np.random.PCG64(1337))
    x0 = np.asarray(pe.x0)[None] + 1e-2 * rng.standard_normal((D, 4))
    kff0 = 1e-4 * rng.standard_normal((D, 1, T))
    s = BatchedILQR(pe, T, np.tile(x0, (reps, 1)), kff_init=np.tile(kff0, (reps, 1, 1)), action_penalty=None,
                    apply_env_kwargs=False)
    traces = s.run(nb_iter=iters)
    xref, uref, ret, lam, st = (s.host(k) for k in ("xref", "uref", "last_return", "lmbda", "status"))
    for arr in (xref, uref, ret, lam, st):
        v = arr.reshape((reps, D) + arr.shape[1:])
        assert np.array_equal(v, np.broadcast_to(v[:1], v.shape)), "copies differ"
    for p in range(0, D, 4):
        r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=iters)
        assert len(traces[p]) == len(r["trace"])
        assert_close(np.array(traces[p]), r["trace"], "trace[%d]" % p)
        assert abs(ret[p] - r["last_return"]) <= TOL * abs(r["last_return"])
        assert st[p] == r["status"] and lam[p] == pytest.approx(r["lmbda"], rel=1e-14)
        assert_close(xref[p], r["xref"], "xref", tol=1e-8)
        assert_close(uref[p], r["uref"], "uref", tol=1e-8)


def _tiled_iteration(oracle, name, kw, N, D, T, iters, alphas, check_every, scale=1e-2, kff_bias=0.0):
    """`N // D` copies of D distinct problems through `iters` full iterations: every copy bitwise the first, the
    distinct ones against the oracle's driver (ilqr.py:136-237) with the same step sizes."""
    from oracle import oracle as O
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    reps = N // D
    pe = envs.make(name, **kw)
    okw = dict(kw)
    okw["uw"] = list(np.asarray(okw["uw"], dtype=float) * np.ones(pe.dm_act))
    oe = O.make_env(name, **okw)
    rng = np.random.Generator(np.random.PCG64(1337))
    x0 = np.asarray(pe.x0)[None] + scale * rng.standard_normal((D, pe.dm_state))
    kff0 = kff_bias + 1e-4 * rng.standard_normal((D, pe.dm_act, T))
    s = BatchedILQR(pe, T, np.tile(x0, (reps, 1)), kff_init=np.tile(kff0, (reps, 1, 1)), action_penalty=None,
                    apply_env_kwargs=False, alphas=alphas)
    traces = s.run(nb_iter=iters)
    xref, uref, ret, lam, st, al = (s.host(k) for k in ("xref", "uref", "last_return", "lmbda", "status", "alpha"))
    for arr in (xref, uref, ret, lam, st, al):
        v = arr.reshape((reps, D) + arr.shape[1:])
        assert np.array_equal(v, np.broadcast_to(v[:1], v.shape), equal_nan=True), "copies differ"
    for p in range(0, D, check_every):
        r = oracle.ilqr_run(oe, T, x0[p], kff0[p], alphas=alphas, nb_iter=iters)
        assert len(traces[p]) == len(r["trace"]), (p, traces[p], r["trace"])
        assert_close(np.array(traces[p]), r["trace"], "trace[%d]" % p)
        assert abs(ret[p] - r["last_return"]) <= TOL * abs(r["last_return"])
        assert st[p] == r["status"] and lam[p] == pytest.approx(r["lmbda"], rel=1e-14)
        assert_close(xref[p], r["xref"], "xref", tol=1e-8)
        assert_close(uref[p], r["uref"], "uref", tol=1e-8)
    return s


def test_ilqr_iteration_full_size_c3(oracle):
    """C3, one GPU's share: 8192 double cartpoles, T = 500, the reference's 11 step sizes; 128 copies of 64."""
    from oracle import oracle as O
    _tiled_iteration(oracle, "double_cartpole", dict(dt=0.01, uw=1e-5), 8192, 64, 500, 2, O.DEFAULT_ALPHAS, 8)


def test_ilqr_iteration_full_size_c4_eight_alphas(oracle):
    """C4 as BASELINE states it: 4096 quadrotors (dx = 12, du = 4), T = 500, the 8-alpha line search
    10**linspace(0,-3,11)[:8]; 128 copies of 32 distinct problems.  The initial feed-forward is the hover
    thrust m g / 4 per rotor (+ the reference's 1e-4 noise, ilqr.py:70): from kff ~ 0 the vehicle free-falls and tumbles
    for the 10 s horizon, a chaotic rollout in which round-off alone decides which step size is accepted."""
    from oracle import oracle as O
    s = _tiled_iteration(oracle, "quadrotor", dict(dt=0.02, uw=1e-3), 4096, 32, 500, 3, O.DEFAULT_ALPHAS[:8], 4,
                         kff_bias=0.5 * 9.81 / 4)
    assert s.alphas.size == 8 and s._cand_state.shape[0] == 17


@pytest.mark.parametrize("sub", [None, 3])
def test_sharded_ilqr_world_size_one_is_the_batched_solver(oracle, sub):
    """ShardedILQR without a process group = BatchedILQR on the whole batch, bit for bit - also when the rank runs its
    block as sub-batches on streams of their own (sub = 3; None: the rule, one batch at this size)."""
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    from trajopt_amd.sharding import ShardedILQR
    N, T, iters = 96, 60, 4
    rng = np.random.Generator(np.random.PCG64(5))
    x0 = np.array([0., np.pi, 0., 0.])[None] + 1e-2 * rng.standard_normal((N, 4))
    kff0 = 1e-4 * rng.standard_normal((N, 1, T))
    sh = ShardedILQR(envs.make("cartpole", dt=0.02), T, x0, kff_init=kff0, action_penalty=1e-5, sub_batches=sub)
    assert 1 <= sh.local.sub_batches <= (sub or 1)
    out = sh.run(nb_iter=iters)
    b = BatchedILQR(envs.make("cartpole", dt=0.02), T, x0, kff_init=kff0, action_penalty=1e-5)
    b.run(nb_iter=iters)
    assert np.array_equal(out["last_return"].cpu().numpy(), b.host("last_return"))
    assert np.array_equal(out["status"].cpu().numpy(), b.host("status"))
    assert np.array_equal(out["iters"].cpu().numpy(), b.host("iters"))
    assert np.array_equal(sh.local.host("xref"), b.host("xref"))
    assert out["progress"]["total_cost"] == pytest.approx(float(b.host("last_return").sum()), rel=1e-12)
    oe = oracle.make_env("cartpole", dt=0.02, uw=[1e-5])
    for p in range(0, N, 19):
        r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=iters)
        assert abs(float(out["last_return"][p]) - r["last_return"]) <= TOL * abs(r["last_return"])


def _sharded_rank(rank, world, port, N, T, iters, q):
    """One rank of the 2-rank rehearsal: both ranks share the box's single GPU, so the process group is gloo
    (RCCL refuses two ranks on one device) and the gathered vectors cross on the host."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from trajopt_amd import envs
    from trajopt_amd.sharding import ShardedILQR
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.Generator(np.random.PCG64(5))
        x0 = np.array([0., np.pi, 0., 0.])[None] + 1e-2 * rng.standard_normal((N, 4))
        kff0 = 1e-4 * rng.standard_normal((N, 1, T))
        sh = ShardedILQR(envs.make("cartpole", dt=0.02), T, x0, kff_init=kff0, action_penalty=1e-5)
        out = sh.run(nb_iter=iters)
        q.put((rank, sh.lo, sh.hi, out["last_return"].cpu().numpy(), out["status"].cpu().numpy(),
               out["iters"].cpu().numpy(), out["progress"]))
    finally:
        dist.destroy_process_group()


def test_sharded_ilqr_two_ranks_equal_one_rank():
    """contiguous block split, no exchange during an iteration, gather at the end.  Two spawned ranks
    (ragged split of 97 problems) must assemble exactly the single-rank result on every rank."""
    import socket
    import torch.multiprocessing as mp
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    N, T, iters, world = 97, 60, 4, 2
    rng = np.random.Generator(np.random.PCG64(5))
    x0 = np.array([0., np.pi, 0., 0.])[None] + 1e-2 * rng.standard_normal((N, 4))
    kff0 = 1e-4 * rng.standard_normal((N, 1, T))
    b = BatchedILQR(envs.make("cartpole", dt=0.02), T, x0, kff_init=kff0, action_penalty=1e-5)
    b.run(nb_iter=iters)
    want = {k: b.host(k).copy() for k in ("last_return", "status", "iters")}
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_rank, args=(r, world, port, N, T, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted((g[1], g[2]) for g in got) == [(0, 49), (49, 97)]
    for rank, lo, hi, ret, st, it, prog in got:
        assert np.array_equal(ret, want["last_return"]) and np.array_equal(st, want["status"])
        assert np.array_equal(it, want["iters"])
        assert prog["total_cost"] == pytest.approx(float(want["last_return"].sum()), rel=1e-12)
        assert prog["nb_stopped"] == int((want["status"] != 0).sum())


@pytest.mark.parametrize("name,kw", [("cartpole", dict(dt=0.02, uw=[1e-5])),
                                     ("pendulum", dict(dt=0.05, uw=[1e-3], slew_rate=True, periodic=True))])
def test_lane_fused_quadratise_backward_at_32768_problems(oracle, name, kw):
    """Batches of >= 32 768 pendulums / cartpoles take quadratisation + backward pass as ONE lane-per-problem launch
    (plan fused_quad_bwd = 2, bwd_tpp.hip): same iterates as the two-launch path on every problem, and as the oracle's
    driver (ilqr.py:136-237) on a sample - with the stopping rules on, so that stopped problems, rejected steps and the
    lambda retry loop all occur inside the batch."""
    from tests.test_ilqr_gpu import _pair
    from trajopt_amd.ilqr import BatchedILQR
    pe, oe = _pair(name, **kw)
    N, T, iters = 32768, 24, 4
    rng = np.random.Generator(np.random.PCG64(11))
    x0 = np.asarray(pe.x0)[None] + 0.3 * rng.standard_normal((N, pe.dm_state))
    kff0 = 1e-2 * rng.standard_normal((N, pe.dm_act, T))
    runs = {}
    for fusion in (True, False):
        s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, no_fusion=not fusion)
        assert s.plan()["fused_quad_bwd"] == (2 if fusion else 0)
        traces = s.run(nb_iter=iters)
        runs[fusion] = dict(traces=traces, **{k: s.host(k) for k in ("xref", "uref", "K", "kff", "lmbda", "dlmbda", "status",
                                                                    "iters", "last_return", "alpha")})
    a, b = runs[True], runs[False]
    assert a["traces"] == b["traces"]
    for k in ("xref", "uref", "K", "kff", "lmbda", "dlmbda", "status", "iters", "last_return"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["alpha"], b["alpha"], equal_nan=True)
    if name == "cartpole":
        assert len({len(t) for t in a["traces"]}) > 1      # the batch really mixes accepting and rejecting problems
    for p in list(range(0, N, 2731)) + [N - 1]:
        r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=iters)
        assert len(a["traces"][p]) == len(r["trace"]), (p, a["traces"][p], r["trace"])
        assert_close(np.array(a["traces"][p]), r["trace"], "trace[%d]" % p)
        assert a["status"][p] == r["status"] and a["iters"][p] == r["iters"]
        assert_close(a["xref"][p], r["xref"], "xref", tol=1e-8)
        if len(r["trace"]) > 1:
            assert_close(a["K"][p], r["K"], "K", tol=1e-8)


@pytest.mark.parametrize("argmin,keep_value", [(False, False), (True, False), (False, True)])
def test_two_pass_line_search_at_16384_problems(oracle, argmin, keep_value):
    """Large batches of small models take the line search in two passes (plan staged_line_search = 2, rollout.hip: the
    returns of every (problem, alpha) without writing a trajectory, then the accepted alpha of each problem rolled out
    again and written): scheduling only - every output must be bitwise what the one-launch line search gives (ilqr.py:
    187-231), on a batch that mixes problems accepting different alphas, problems rejecting all of them and problems
    that stop (tolfun) at different iterations; a sample against the oracle's driver."""
    from tests.test_ilqr_gpu import _pair
    from trajopt_amd.ilqr import BatchedILQR
    pe, oe = _pair("cartpole", dt=0.02, uw=[1e-5])
    N, T = 16384, 24
    rng = np.random.Generator(np.random.PCG64(12))
    x0 = np.asarray(pe.x0)[None] + 0.3 * rng.standard_normal((N, pe.dm_state))
    kff0 = 1e-2 * rng.standard_normal((N, pe.dm_act, T))
    keys = ("xref", "uref", "K", "kff", "lmbda", "dlmbda", "status", "iters", "last_return", "alpha") + (("V", "v", "Quu") if keep_value else ())
    runs = {}
    for mode in (-1, 0):
        s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, staged_line_search=mode,
                        argmin_alpha=argmin, keep_value=keep_value, tolfun=100.0)
        assert s.plan()["staged_line_search"] == (2 if mode < 0 else 0)
        s.run(nb_iter=4)                    # iterations chained in one call: problems that stop early keep their
        s.iterate(1)                        # reference in a plane the later iterations write around; and single ones
        traces = s.run(nb_iter=1)
        runs[mode] = dict(traces=traces, **{k: s.host(k) for k in keys})
    a, b = runs[-1], runs[0]
    assert a["traces"] == b["traces"]
    for k in keys:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert len(np.unique(a["alpha"][~np.isnan(a["alpha"])])) > 1, "the batch must accept different step sizes"
    assert len({len(t) for t in a["traces"]}) > 1, "the batch must mix accepting and rejecting problems"
    assert len(np.unique(a["iters"][a["status"] == 2])) > 1 and (a["status"] == 0).any(), "problems must stop at different iterations"
    if not argmin and not keep_value:
        for p in list(range(0, N, 2731)) + [N - 1]:
            r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=6, tolfun=100.0)
            assert a["status"][p] == r["status"] and a["iters"][p] == r["iters"]
            assert abs(a["last_return"][p] - r["last_return"]) <= TOL * abs(r["last_return"])
            assert_close(a["xref"][p], r["xref"], "xref", tol=1e-8)


def test_two_pass_line_search_falls_back_where_there_is_no_returns_only_rollout():
    """Asked for explicitly on an env the returns-only rollout is not built for (slew-rate cost, wrapped angle), or
    with a single alpha, the plan says one launch - and the solver runs."""
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    rng = np.random.Generator(np.random.PCG64(13))
    N, T = 256, 16
    for name, kw, extra in (("pendulum", dict(dt=0.05, slew_rate=True, periodic=True), {}),
                            ("cartpole", dict(dt=0.02), dict(alphas=np.array([1.0])))):
        env = envs.make(name, **kw)
        x0 = np.asarray(env.x0)[None] + 0.1 * rng.standard_normal((N, env.dm_state))
        s = BatchedILQR(env, T, x0, staged_line_search=2, **extra)
        assert s.plan()["staged_line_search"] == 0
        s.run(nb_iter=2)
        assert np.all(np.isfinite(s.host("last_return")))


_NCCL_RANK = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from trajopt_amd import envs
from trajopt_amd.ilqr import BatchedILQR
from trajopt_amd.sharding import ShardedILQR
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
N, T, iters = 96, 60, 4
rng = np.random.Generator(np.random.PCG64(5))
x0 = np.array([0., np.pi, 0., 0.])[None] + 1e-2 * rng.standard_normal((N, 4))
kff0 = 1e-4 * rng.standard_normal((N, 1, T))
sh = ShardedILQR(envs.make("cartpole", dt=0.02), T, x0, kff_init=kff0, action_penalty=1e-5)
out = sh.run(nb_iter=iters)
assert out["last_return"].is_cuda and out["status"].is_cuda, "RCCL gathers device tensors in place"
b = BatchedILQR(envs.make("cartpole", dt=0.02), T, x0, kff_init=kff0, action_penalty=1e-5)
b.run(nb_iter=iters)
ok = (np.array_equal(out["last_return"].cpu().numpy(), b.host("last_return")) and
      np.array_equal(out["status"].cpu().numpy(), b.host("status")) and np.array_equal(out["iters"].cpu().numpy(), b.host("iters")))
tt = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dist.barrier()
print(json.dumps({"ok": bool(ok), "backend": dist.get_backend(), "max": float(tt.item()),
                  "total": out["progress"]["total_cost"], "want_total": float(b.host("last_return").sum())}))
dist.destroy_process_group()
"""


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def test_rccl_process_group_of_one_rank_on_the_device():
    """the `nccl` (= RCCL) branch of the sharded solver on the real device - a one-rank process
    group, device tensors through all_gather / all_reduce / barrier (mistakes a gloo rehearsal cannot see) - in a child
    process so that this one keeps no process group."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _NCCL_RANK, root, str(_free_port())], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ok"] and d["backend"] == "nccl" and d["max"] == 1.25
    assert d["total"] == pytest.approx(d["want_total"], rel=1e-12)


def test_bench_collective_path_with_rccl_at_world_size_one():
    """bench.py's multi-rank path (RCCL barrier, max over ranks on a device tensor, one device per rank checked) with
    ONE rank on the device: what the driver's N > 1 runs execute, minus the other ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRAJOPT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--instances", "512",
                        "--horizon", "48", "--no-configs", "--no-cpu-baseline", "--sub-batches", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and len(d["rank_devices"]) == 1 and d["value"] > 0
    # (--sub-batches 2: the timed run is the sub-batched one - side streams next to RCCL's own - and the one-batch time rides along)
    assert d["sub_batches"] in (1, 2) and d["one_batch_ms_per_step"] > 0
