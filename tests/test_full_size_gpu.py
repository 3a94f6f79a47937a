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
