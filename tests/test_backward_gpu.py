"""GPU parity of the HIP backward passes (rows A1, A9) against the CPU oracle, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from tests.util import TOL, assert_close, rel_err

pytestmark = pytest.mark.gpu

OUT = ("Qxx", "Qux", "Quu", "qx", "qu", "V", "v", "dV", "K", "kff")


def _run_gpu(stack, lmbda, reg, n, m, T, N, layout, full=True, retry=False, sched=None, uref=None, kernel=0):
    import torch
    from trajopt_amd import backward, layout as L
    d = {k: L.to_device(v, layout) for k, v in stack.items()}
    lam = torch.full((N,), float(lmbda), dtype=torch.float64, device="cuda") if np.isscalar(lmbda) else \
        torch.tensor(lmbda, dtype=torch.float64, device="cuda")
    dl = torch.ones(N, dtype=torch.float64, device="cuda") if retry else None
    u = L.to_device(uref, layout) if uref is not None else None
    o = backward.ilqr_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], lam, reg, n, m, T,
                                    N, layout, full=full, retry=retry, sched=sched, dlmbda=dl, uref=u, kernel=kernel)
    torch.cuda.synchronize()
    res = {k: L.from_device(o[k], layout) for k in o if k not in ("diverge", "g_norm")}
    res["diverge"] = o["diverge"].cpu().numpy()
    if "g_norm" in o:
        res["g_norm"] = o["g_norm"].cpu().numpy()
    res["lmbda"] = lam.cpu().numpy()
    if dl is not None:
        res["dlmbda"] = dl.cpu().numpy()
    return res


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("n,m,T,N", [(2, 1, 100, 3), (3, 1, 33, 65), (4, 1, 64, 130), (4, 2, 40, 67), (6, 1, 48, 70),
                                      (12, 4, 32, 5), (5, 3, 20, 4), (64, 16, 6, 2)])
@pytest.mark.parametrize("reg", [1, 2])
def test_ilqr_backward_matches_oracle(oracle, n, m, T, N, layout, reg):
    from trajopt_amd import synthetic as S
    stack = S.backward_inputs(n, m, T, N, seed=1337 + n)
    res = _run_gpu(stack, 1.0, reg, n, m, T, N, layout)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 1.0, reg, n,
                                           m, T)
        assert not failed and res["diverge"][p] == -1
        for k, r in zip(OUT, ref[:10]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))


# kernel families (trajopt_backward_io_t::kernel, include/trajopt_amd.h): 1 workgroup-per-problem, 2 MFMA
# wave-per-problem, 3 lane-per-problem, 4 MFMA four-problems-per-wave, 5 column-per-lane (eight lanes per problem)
# (0 = the library's own choice)


@pytest.mark.parametrize("kernel,n,m,T,N,layout", [
    (1, 4, 1, 40, 9, 1), (1, 6, 1, 30, 5, 0), (1, 12, 4, 20, 3, 1), (1, 12, 4, 20, 3, 0),      # workgroup = problem
    (2, 4, 1, 40, 9, 1), (2, 4, 1, 40, 9, 0), (2, 6, 1, 50, 7, 0), (2, 6, 1, 50, 7, 1),        # wave = problem (MFMA)
    (2, 12, 4, 40, 6, 0), (2, 12, 4, 40, 6, 1), (2, 8, 2, 25, 4, 0), (2, 13, 3, 12, 3, 0), (2, 15, 1, 12, 3, 1),
    (2, 3, 2, 12, 3, 0), (2, 9, 4, 12, 3, 0), (2, 2, 1, 12, 3, 1),
    (3, 6, 1, 50, 70, 1), (3, 4, 2, 30, 66, 1), (3, 2, 1, 30, 5, 1),                            # lane = problem
    (4, 4, 1, 64, 130, 1), (4, 4, 1, 40, 1, 0), (4, 3, 1, 33, 21, 1), (4, 2, 1, 100, 3, 1),    # MFMA, 4 problems / wave
    (4, 4, 1, 500, 37, 1),
    (5, 6, 1, 50, 70, 1), (5, 6, 1, 33, 7, 1), (5, 6, 1, 500, 37, 1), (5, 6, 1, 21, 1, 0),    # eight lanes per problem
    (5, 6, 1, 1, 3, 1), (5, 6, 1, 2, 9, 1), (5, 6, 1, 3, 8, 1), (3, 6, 1, 1, 3, 1),
])
@pytest.mark.parametrize("reg", [1, 2])
def test_every_kernel_family_matches_oracle(oracle, kernel, n, m, T, N, layout, reg):
    from trajopt_amd import synthetic as S
    stack = S.backward_inputs(n, m, T, N, seed=4000 + 10 * n + m)
    uref = np.random.default_rng(n).standard_normal((N, m, T))
    res = _run_gpu(stack, 0.8, reg, n, m, T, N, layout, uref=uref, kernel=kernel)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 0.8, reg, n,
                                           m, T)
        assert not failed and res["diverge"][p] == -1
        for k, r in zip(OUT, ref[:10]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))
        g = np.mean(np.max(np.abs(ref[9]) / (np.abs(uref[p]) + 1.0), axis=1))
        assert res["g_norm"][p] == pytest.approx(g, rel=1e-9)


@pytest.mark.parametrize("kernel,n,m", [(1, 6, 2), (2, 6, 2), (3, 4, 1), (4, 4, 1), (4, 3, 1), (5, 6, 1)])
def test_asymmetric_cost_hessians_are_taken_as_given(oracle, kernel, n, m):
    """util.cpp uses Cxx as passed (no symmetry assumed); only V is symmetrised (:161)."""
    from trajopt_amd import synthetic as S
    T, N = 10, 3
    stack = S.backward_inputs(n, m, T, N, seed=99)
    Cxx = np.array(stack["Cxx"])
    Cxx[:, 0, 1, :] += 0.05          # asymmetric, terminal slice included
    stack = dict(stack, Cxx=Cxx)
    res = _run_gpu(stack, 1.0, 1, n, m, T, N, 1 if kernel >= 3 else 0, kernel=kernel)
    for p in range(N):
        q = S.problem(stack, p)
        ref, _ = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 1.0, 1, n, m, T)
        for k, r in zip(OUT, ref[:10]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("n,m,T,N", [(4, 1, 500, 64), (6, 1, 500, 64)])
def test_gains_only_long_horizon(oracle, n, m, T, N, layout):
    """BASELINE shapes C2/C3 at full T; gains-only outputs (what the fused iteration uses)."""
    from trajopt_amd import synthetic as S
    stack = S.backward_inputs(n, m, T, N)
    res = _run_gpu(stack, 1.0, 1, n, m, T, N, layout, full=False)
    assert "V" not in res
    for p in range(0, N, 7):
        q = S.problem(stack, p)
        ref, _ = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 1.0, 1, n, m, T)
        assert_close(res["K"][p], ref[8], "K")
        assert_close(res["kff"][p], ref[9], "kff")
        assert_close(res["dV"][p], ref[7], "dV")


def _indefinite(stack, at, N):
    """Make Cuu strongly negative at time index `at` for odd problems."""
    Cuu = np.array(stack["Cuu"])
    m = Cuu.shape[1]
    for p in range(1, N, 2):
        Cuu[p, :, :, at] = -50.0 * np.eye(m)
    out = dict(stack)
    out["Cuu"] = Cuu
    return out


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("n,m", [(2, 1), (4, 1), (4, 2), (12, 4)])
def test_divergence_index_and_partial_outputs(oracle, n, m, layout):
    from trajopt_amd import synthetic as S
    T, N, at = 24, 6, 9
    stack = _indefinite(S.backward_inputs(n, m, T, N, seed=5), at, N)
    res = _run_gpu(stack, 1e-3, 1, n, m, T, N, layout)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 1e-3, 1, n,
                                           m, T)
        if p % 2 == 1:
            assert failed and ref[10] == at and res["diverge"][p] == at
            # slices above the failing step are valid; the Q blocks OF the failing step are written too
            assert_close(res["V"][p][..., at + 1:], ref[5][..., at + 1:], "V")
            assert_close(res["K"][p][..., at + 1:], ref[8][..., at + 1:], "K")
            assert_close(res["Qxx"][p][..., at:], ref[0][..., at:], "Qxx")
            assert_close(res["Quu"][p][..., at:], ref[2][..., at:], "Quu")
            # nothing below the failing step is touched (the wrapper zero-fills like the oracle)
            assert not np.any(res["K"][p][..., :at + 1]) and not np.any(res["V"][p][..., :at + 1])
        else:
            assert not failed and res["diverge"][p] == -1
            assert_close(res["K"][p], ref[8], "K")


@pytest.mark.parametrize("kernel", [3, 5])
@pytest.mark.parametrize("reg", [1, 2])
def test_dx6_families_divergence_partial_outputs_and_retry(oracle, kernel, reg):
    """The two struct-of-arrays families for dx = 6 (lane = problem, eight lanes per problem; the dispatcher picks by
    batch size): divergence index, outputs kept above the failing step, untouched below it, and the folded lambda
    retry loop - with diverging and healthy problems mixed inside one wave."""
    from trajopt_amd import synthetic as S
    from trajopt_amd._lib import LambdaSchedule
    n, m, T, N, at = 6, 1, 19, 13, 7
    stack = _indefinite(S.backward_inputs(n, m, T, N, seed=21), at, N)
    res = _run_gpu(stack, 1e-3, reg, n, m, T, N, 1, kernel=kernel)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 1e-3, reg, n, m, T)
        if p % 2 == 1:
            assert failed and ref[10] == at and res["diverge"][p] == at
            for k, i in (("V", 5), ("v", 6), ("K", 8), ("kff", 9)):
                assert_close(res[k][p][..., at + 1:], ref[i][..., at + 1:], k)
            for k, i in (("Qxx", 0), ("Qux", 1), ("Quu", 2), ("qx", 3), ("qu", 4)):
                assert_close(res[k][p][..., at:], ref[i][..., at:], k)
            assert_close(res["dV"][p], ref[7], "dV")
            assert not np.any(res["K"][p][..., :at + 1]) and not np.any(res["V"][p][..., :at + 1])
            assert not np.any(res["Qxx"][p][..., :at])
        else:
            assert not failed and res["diverge"][p] == -1
            for k, r in zip(OUT, ref[:10]):
                assert_close(res[k][p], r, "%s[p=%d]" % (k, p))
    sched = LambdaSchedule(1e-6, 1e6, 1.6)
    res = _run_gpu(stack, 1.0, reg, n, m, T, N, 1, retry=True, sched=sched, kernel=kernel)
    for p in range(N):
        q = S.problem(stack, p)
        lam, dlam = 1.0, 1.0
        while True:
            ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], lam, reg, n, m, T)
            if not failed:
                break
            dlam = max(dlam * 1.6, 1.6)
            lam = max(lam * dlam, 1e-6)
            if lam > 1e6:
                break
        assert res["lmbda"][p] == pytest.approx(lam, rel=1e-15) and res["dlmbda"][p] == pytest.approx(dlam, rel=1e-15)
        assert (res["diverge"][p] >= 0) == failed
        if not failed:
            for k, r in zip(OUT, ref[:10]):
                assert_close(res[k][p], r, "%s[p=%d] after retries" % (k, p))


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("n,m", [(4, 1), (12, 4)])
def test_in_kernel_lambda_retry_matches_driver_loop(oracle, n, m, layout):
    """ilqr.py:162-173 folded into the kernel: same final lmbda/dlmbda and gains as looping the oracle."""
    from trajopt_amd import synthetic as S
    from trajopt_amd._lib import LambdaSchedule
    T, N, at = 16, 5, 4
    stack = _indefinite(S.backward_inputs(n, m, T, N, seed=11), at, N)
    sched = LambdaSchedule(1e-6, 1e6, 1.6)
    res = _run_gpu(stack, 1.0, 1, n, m, T, N, layout, retry=True, sched=sched)
    for p in range(N):
        q = S.problem(stack, p)
        lam, dlam = 1.0, 1.0
        while True:
            ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], lam, 1,
                                               n, m, T)
            if not failed:
                break
            dlam = max(dlam * 1.6, 1.6)
            lam = max(lam * dlam, 1e-6)
            if lam > 1e6:
                break
        assert res["lmbda"][p] == pytest.approx(lam, rel=1e-15)
        assert res["dlmbda"][p] == pytest.approx(dlam, rel=1e-15)
        assert (res["diverge"][p] >= 0) == failed
        if not failed:
            assert_close(res["K"][p], ref[8], "K")
            assert_close(res["kff"][p], ref[9], "kff")
            assert_close(res["dV"][p], ref[7], "dV")
    # lambda above max: gives up with the flag set (ilqr.py:168)
    sched2 = LambdaSchedule(1e-6, 10.0, 1.6)
    res2 = _run_gpu(stack, 1.0, 1, n, m, T, N, layout, retry=True, sched=sched2)
    assert all(res2["diverge"][p] == at for p in range(1, N, 2))
    assert all(res2["lmbda"][p] > 10.0 for p in range(1, N, 2))


@pytest.mark.parametrize("layout", [0, 1])
def test_gradient_norm_output(oracle, layout):
    from trajopt_amd import synthetic as S
    n, m, T, N = 4, 2, 20, 9
    stack = S.backward_inputs(n, m, T, N, seed=3)
    uref = np.random.default_rng(0).standard_normal((N, m, T))
    res = _run_gpu(stack, 1.0, 1, n, m, T, N, layout, full=False, uref=uref)
    for p in range(N):
        g = np.mean(np.max(np.abs(res["kff"][p]) / (np.abs(uref[p]) + 1.0), axis=1))  # ilqr.py:176
        assert res["g_norm"][p] == pytest.approx(g, rel=1e-14)


def test_host_entry_point_is_the_reference_signature(oracle):
    """trajopt.ilqr.core.backward_pass drop-in: host arrays in, 11-tuple of F-ordered arrays out."""
    from trajopt_amd import synthetic as S
    from trajopt_amd.ilqr.core import backward_pass
    for (n, m, T) in [(2, 1, 100), (4, 1, 50), (12, 4, 20)]:
        q = S.problem(S.backward_inputs(n, m, T, 1), 0)
        # C-ordered inputs, as the reference's Python side allocates them (np.zeros((n, n, T)))
        args = [np.ascontiguousarray(q[k]) for k in ("Cxx", "cx", "Cuu", "cu", "Cxu", "A", "B")]
        out = backward_pass(*args, 1.0, 1, n, m, T)
        ref, _ = oracle.backward_pass(*args, 1.0, 1, n, m, T)
        assert len(out) == 11 and out[10] == 0
        for k, a, r in zip(OUT, out[:10], ref[:10]):
            assert a.flags.f_contiguous and a.dtype == np.float64
            assert_close(a, r, k)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("n,m,T,N", [(2, 1, 60, 3), (4, 1, 40, 5), (6, 2, 30, 4), (12, 4, 24, 3), (64, 16, 5, 2),
                                      (17, 1, 9, 3), (33, 7, 9, 3), (48, 16, 7, 2), (64, 16, 30, 3), (50, 3, 11, 2)])
def test_riccati_backward_matches_oracle(oracle, n, m, T, N, layout):
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    stack = S.backward_inputs(n, m, T, N, seed=77, affine=True)
    d = {k: L.to_device(v, layout) for k, v in stack.items()}
    o = backward.riccati_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], d["c"], n, m, T,
                                       N, layout)
    torch.cuda.synchronize()
    assert not o["status"].cpu().numpy().any()
    for p in range(N):
        q = S.problem(stack, p)
        K, kff, V, v = oracle.riccati_backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"],
                                                    q["c"], n, m, T)
        assert_close(L.from_device(o["K"], layout)[p], K, "K")
        assert_close(L.from_device(o["kff"], layout)[p], kff, "kff")
        assert_close(L.from_device(o["V"], layout)[p], V, "V")
        assert_close(L.from_device(o["v"], layout)[p], v, "v")


@pytest.mark.parametrize("kernel", [0, 3, 4, 5])
@pytest.mark.parametrize("n,m,T,N", [(4, 1, 64, 130), (2, 1, 33, 5), (3, 1, 20, 17), (6, 1, 30, 70), (4, 2, 25, 66)])
@pytest.mark.parametrize("reg", [1, 2])
def test_structured_cost_matches_dense_oracle(oracle, kernel, n, m, T, N, reg):
    """trajopt_backward_io_t::cost_w: Cxx = w[t] diag(gxx), Cuu = diag(guu), Cxu = 0 formed inside the kernel give
    what the oracle gives on the equivalent dense arrays (both kernel families that implement it)."""
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    if kernel == 4 and not (m == 1 and n <= 4):
        pytest.skip("the four-problems-per-wave kernel covers dm_state <= 4, dm_act = 1")
    if kernel == 5 and not (m == 1 and n == 6):
        pytest.skip("the column-per-lane kernel covers dm_state = 6, dm_act = 1")
    rng = np.random.default_rng(100 * n + m)
    stack = S.backward_inputs(n, m, T, N, seed=900 + n)
    w = np.abs(rng.standard_normal(T + 1)) + 0.1
    w[T // 2] = 0.0                                   # `if a:` drops the state cost for a zero weight
    gxx, guu = np.abs(rng.standard_normal(n)) + 0.5, np.abs(rng.standard_normal(m)) + 0.05
    Cxx = np.zeros((N, n, n, T + 1)); Cuu = np.zeros((N, m, m, T + 1)); Cxu = np.zeros((N, n, m, T + 1))
    Cxx[:, np.arange(n), np.arange(n), :] = gxx[None, :, None] * w[None, None, :]
    Cuu[:, np.arange(m), np.arange(m), :] = guu[None, :, None]
    stack = dict(stack, Cxx=Cxx, Cuu=Cuu, Cxu=Cxu)
    d = {k: L.to_device(stack[k], 1) for k in ("cx", "cu", "A", "B")}
    uref = rng.standard_normal((N, m, T))
    lam = torch.full((N,), 0.7, dtype=torch.float64, device="cuda")
    o = backward.ilqr_backward_pass(None, d["cx"], None, d["cu"], None, d["A"], d["B"], lam, reg, n, m, T, N, 1,
                                    uref=L.to_device(uref, 1), kernel=kernel,
                                    cost=(torch.as_tensor(w, device="cuda"), gxx, guu))
    torch.cuda.synchronize()
    assert not (o["diverge"].cpu().numpy() >= 0).any()
    for p in range(0, N, 3):
        q = S.problem(stack, p)
        ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 0.7, reg,
                                           n, m, T)
        assert not failed
        for k, r in zip(OUT, ref[:10]):
            assert_close(L.from_device(o[k], 1)[p], r, "%s[p=%d]" % (k, p))
        g = np.mean(np.max(np.abs(ref[9]) / (np.abs(uref[p]) + 1.0), axis=1))
        assert o["g_norm"][p].item() == pytest.approx(g, rel=1e-9)


def test_structured_cost_is_refused_where_no_kernel_takes_it():
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    from trajopt_amd._lib import TrajoptError, lib
    n, m, T, N = 20, 5, 8, 3        # workgroup-per-problem kernel: dense cost arrays only
    assert lib().trajopt_backward_supports_structured_cost(n, m, T, C.c_int64(N), 1) == 0
    assert lib().trajopt_backward_supports_structured_cost(4, 1, 500, C.c_int64(4096), 1) == 1
    assert lib().trajopt_backward_supports_structured_cost(4, 1, 500, C.c_int64(4096), 0) == 1    # matrix-core kernel
    assert lib().trajopt_backward_supports_structured_cost(12, 4, 500, C.c_int64(4096), 0) == 1
    stack = S.backward_inputs(n, m, T, N, seed=1)
    d = {k: L.to_device(stack[k], 1) for k in ("cx", "cu", "A", "B")}
    lam = torch.ones(N, dtype=torch.float64, device="cuda")
    with pytest.raises(TrajoptError):
        backward.ilqr_backward_pass(None, d["cx"], None, d["cu"], None, d["A"], d["B"], lam, 1, n, m, T, N, 1,
                                    cost=(torch.ones(T + 1, dtype=torch.float64, device="cuda"), np.ones(n), np.ones(m)))
    # a forced family that does not take it is refused as well (the workgroup kernel for a shape the matrix cores take)
    n, m = 12, 4
    stack = S.backward_inputs(n, m, T, N, seed=1)
    d = {k: L.to_device(stack[k], 0) for k in ("cx", "cu", "A", "B")}
    with pytest.raises(TrajoptError):
        backward.ilqr_backward_pass(None, d["cx"], None, d["cu"], None, d["A"], d["B"], lam, 1, n, m, T, N, 0, kernel=1,
                                    cost=(torch.ones(T + 1, dtype=torch.float64, device="cuda"), np.ones(n), np.ones(m)))


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("n,m,T,N", [(12, 4, 40, 9), (5, 2, 33, 7), (8, 3, 21, 5), (11, 2, 18, 4), (4, 1, 30, 6)])
@pytest.mark.parametrize("reg", [1, 2])
@pytest.mark.parametrize("full", [True, False])
def test_structured_cost_on_the_matrix_core_kernel_matches_dense_oracle(oracle, layout, n, m, T, N, reg, full):
    """The wave-per-problem matrix-core kernel (C4's family, forced: kernel = 2) with the structured cost, both layouts,
    both regularisation modes (reg = 1 is a compile-time instantiation of its own), full and gains-only outputs, a zero
    weight (`if a:` drops the state cost) - against the oracle on the equivalent dense arrays."""
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    rng = np.random.default_rng(100 * n + m)
    stack = S.backward_inputs(n, m, T, N, seed=700 + n)
    w = np.abs(rng.standard_normal(T + 1)) + 0.1
    w[T // 2] = 0.0
    gxx, guu = np.abs(rng.standard_normal(n)) + 0.5, np.abs(rng.standard_normal(m)) + 0.05
    Cxx = np.zeros((N, n, n, T + 1)); Cuu = np.zeros((N, m, m, T + 1)); Cxu = np.zeros((N, n, m, T + 1))
    Cxx[:, np.arange(n), np.arange(n), :] = gxx[None, :, None] * w[None, None, :]
    Cuu[:, np.arange(m), np.arange(m), :] = guu[None, :, None]
    stack = dict(stack, Cxx=Cxx, Cuu=Cuu, Cxu=Cxu)
    d = {k: L.to_device(stack[k], layout) for k in ("cx", "cu", "A", "B")}
    lam = torch.full((N,), 0.7, dtype=torch.float64, device="cuda")
    o = backward.ilqr_backward_pass(None, d["cx"], None, d["cu"], None, d["A"], d["B"], lam, reg, n, m, T, N, layout,
                                    kernel=2, full=full, cost=(torch.as_tensor(w, device="cuda"), gxx, guu))
    torch.cuda.synchronize()
    assert not (o["diverge"].cpu().numpy() >= 0).any()
    keys = OUT if full else ("dV", "K", "kff")
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 0.7, reg, n, m, T)
        assert not failed
        for k, r in zip(OUT, ref[:10]):
            if k in keys:
                assert_close(L.from_device(o[k], layout)[p], r, "%s[p=%d]" % (k, p))


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("n,m,max_lmbda", [(12, 4, 1e6), (12, 4, 300.0), (5, 2, 1e6)])
def test_lambda_ladder_in_three_launches_matches_the_sequential_loop(oracle, layout, n, m, max_lmbda):
    """The driver's configuration of the matrix-core kernel (gains only, reg = 1, structured cost, retries on) runs the
    lambda ladder of ilqr.py:162-173 in three launches: the first rung (and TRAJOPT_LADDER_INKERNEL retries) inside the
    wave, the rungs beyond probed in parallel, then a resume at the rung the loop would have stopped at (bwd_mfma16.hip).  A negative control cost makes
    Quu indefinite until lambda exceeds ~|guu|, and lambda starts anywhere between 1e-4 and 1e2: problems that need no
    retry, a few (found by a probe), more than the probed rungs (resume and go on
    alone) and - with a low max_lmbda - problems that give up at every stage.  Final lambda / dlambda, the failure
    flag and index, gains and dV must be those of looping the oracle."""
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    from trajopt_amd._lib import LambdaSchedule
    T, N = 24, 48
    rng = np.random.default_rng(5 * n + m)
    stack = S.backward_inputs(n, m, T, N, seed=900 + n)
    w = np.abs(rng.standard_normal(T + 1)) + 0.1
    gxx, guu = np.abs(rng.standard_normal(n)) + 0.5, -(60.0 + 40.0 * rng.random(m))
    Cxx = np.zeros((N, n, n, T + 1)); Cuu = np.zeros((N, m, m, T + 1)); Cxu = np.zeros((N, n, m, T + 1))
    Cxx[:, np.arange(n), np.arange(n), :] = gxx[None, :, None] * w[None, None, :]
    Cuu[:, np.arange(m), np.arange(m), :] = guu[None, :, None]
    stack = dict(stack, Cxx=Cxx, Cuu=Cuu, Cxu=Cxu)
    d = {k: L.to_device(stack[k], layout) for k in ("cx", "cu", "A", "B")}
    lam0 = 10.0 ** rng.uniform(-4.0, 2.3, N)
    lam0[:4] = [1e-4, 1e-3, 150.0, 250.0]
    lam = torch.tensor(lam0, dtype=torch.float64, device="cuda")
    dl = torch.ones(N, dtype=torch.float64, device="cuda")
    sched = LambdaSchedule(1e-6, max_lmbda, 1.6)
    o = backward.ilqr_backward_pass(None, d["cx"], None, d["cu"], None, d["A"], d["B"], lam, 1, n, m, T, N, layout,
                                    kernel=2, full=False, retry=True, sched=sched, dlmbda=dl,
                                    cost=(torch.as_tensor(w, device="cuda"), gxx, guu))
    torch.cuda.synchronize()
    div = o["diverge"].cpu().numpy()
    lam1, dl1 = lam.cpu().numpy(), dl.cpu().numpy()
    rungs = []
    for p in range(N):
        q = S.problem(stack, p)
        l, dlm, k = float(lam0[p]), 1.0, 0
        while True:
            ref, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], l, 1, n, m, T)
            if not failed:
                break
            dlm = max(dlm * 1.6, 1.6)
            l = max(l * dlm, 1e-6)
            if l > max_lmbda:
                break
            k += 1
        rungs.append(k if not failed else -1)
        assert lam1[p] == pytest.approx(l, rel=1e-15), (p, k)
        assert dl1[p] == pytest.approx(dlm, rel=1e-15), (p, k)
        assert (div[p] >= 0) == bool(failed), (p, k, div[p])
        if failed:
            assert div[p] == ref[10], (p, k, div[p], ref[10])   # the step the last attempted sweep failed at
        else:
            assert div[p] == -1
            assert_close(L.from_device(o["K"], layout)[p], ref[8], "K[p=%d, rung %d]" % (p, k))
            assert_close(L.from_device(o["kff"], layout)[p], ref[9], "kff[p=%d, rung %d]" % (p, k))
            assert_close(L.from_device(o["dV"], layout)[p], ref[7], "dV[p=%d, rung %d]" % (p, k))
    rungs = np.array(rungs)
    # the batch must exercise every stage of the ladder
    assert (rungs == 0).any() and ((rungs >= 1) & (rungs <= 2)).any() and (rungs >= 3).any(), rungs
    if max_lmbda < 1e6:
        assert (rungs == -1).any(), rungs


def test_a_forced_kernel_family_without_the_shape_is_refused():
    """trajopt_backward_io_t::kernel names a family per call (no global state): where that family has no kernel for
    the shape or layout the call fails with TRAJOPT_ERR_UNSUPPORTED instead of silently taking another one."""
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    from trajopt_amd._lib import TrajoptError
    n, m, T, N = 6, 2, 8, 3
    stack = S.backward_inputs(n, m, T, N, seed=1)
    for kernel, layout in ((4, 1), (3, 1), (3, 0), (7, 1)):
        d = {k: L.to_device(v, layout) for k, v in stack.items()}
        lam = torch.ones(N, dtype=torch.float64, device="cuda")
        with pytest.raises(TrajoptError):
            backward.ilqr_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], lam, 1, n, m, T,
                                        N, layout, kernel=kernel)


@pytest.mark.parametrize("n,m,T,N", [(33, 7, 9, 4), (64, 16, 6, 4)])
def test_riccati_sweep_does_not_depend_on_the_kernel_for_asymmetric_or_indefinite_inputs(oracle, n, m, T, N):
    """The tiled matrix-core kernel (16 < dx <= 64) assumes a symmetric Cxx and a positive definite Quu; the reference
    (riccati.py:100-126, np.linalg.inv) assumes neither.  Problems that break either assumption must come out as the
    generic kernel / the oracle computes them, the others untouched."""
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    stack = S.backward_inputs(n, m, T, N, seed=31, affine=True)
    rng = np.random.default_rng(8)
    stack["Cxx"][1][:, :, T // 2] += 0.3 * rng.standard_normal((n, n))   # problem 1: one asymmetric slice
    stack["Cuu"][2][...] = -stack["Cuu"][2] - 5.0 * np.eye(m)[:, :, None]  # problem 2: Quu indefinite (negative definite) but invertible
    d = {k: L.to_device(v, 0) for k, v in stack.items()}
    o = backward.riccati_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], d["c"], n, m, T, N, 0)
    torch.cuda.synchronize()
    assert not o["status"].cpu().numpy().any()
    for p in range(N):
        q = S.problem(stack, p)
        K, kff, V, v = oracle.riccati_backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"],
                                                    q["c"], n, m, T)
        assert_close(L.from_device(o["K"], 0)[p], K, "K of problem %d" % p)
        assert_close(L.from_device(o["V"], 0)[p], V, "V of problem %d" % p)
        assert_close(L.from_device(o["v"], 0)[p], v, "v of problem %d" % p)


def test_riccati_sweeps_overlapping_on_two_streams_keep_their_own_fallback_flags(oracle):
    """the matrix-core sweep hands a problem it cannot take (here: an indefinite Quu) to the generic kernel
    through a per-problem flag.  That flag lives in the caller's status array, so two calls that overlap on one GPU -
    one batch with such a problem, one without, each on its own stream, several rounds - cannot clear or steal each
    other's flags; a call WITHOUT a status array (the per-device static flags) gives the same results on its own."""
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    n, m, T, N = 33, 7, 40, 24
    stacks = [S.backward_inputs(n, m, T, N, seed=41, affine=True), S.backward_inputs(n, m, T, N, seed=42, affine=True)]
    for p in (3, 17):
        stacks[0]["Cuu"][p][...] = -stacks[0]["Cuu"][p] - 5.0 * np.eye(m)[:, :, None]   # indefinite but invertible
    dev = [{k: L.to_device(v, 0) for k, v in st.items()} for st in stacks]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = [[], []]
    for rnd in range(6):
        for i in (rnd % 2, 1 - rnd % 2):   # alternate who goes first
            with torch.cuda.stream(streams[i]):
                d = dev[i]
                outs[i].append(backward.riccati_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"],
                                                              d["c"], n, m, T, N, 0))
    torch.cuda.synchronize()
    refs = []
    for i in range(2):
        refs.append({})
        for p in (0, 3, 17, N - 1):
            q = S.problem(stacks[i], p)
            refs[i][p] = oracle.riccati_backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], q["c"], n, m, T)
    for i in range(2):
        for o in outs[i]:
            assert not o["status"].cpu().numpy().any()
            assert torch.equal(o["K"], outs[i][0]["K"]) and torch.equal(o["V"], outs[i][0]["V"])   # every round the same bits
        for p, (K, kff, V, v) in refs[i].items():
            assert_close(L.from_device(outs[i][0]["K"], 0)[p], K, "K of batch %d problem %d" % (i, p))
            assert_close(L.from_device(outs[i][0]["V"], 0)[p], V, "V of batch %d problem %d" % (i, p))
            assert_close(L.from_device(outs[i][0]["kff"], 0)[p], kff, "kff of batch %d problem %d" % (i, p))
    d = dev[0]
    o = backward.riccati_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], d["c"], n, m, T, N, 0,
                                       with_status=False)
    torch.cuda.synchronize()
    assert o["status"] is None
    assert torch.equal(o["K"], outs[0][0]["K"]) and torch.equal(o["V"], outs[0][0]["V"]) and torch.equal(o["v"], outs[0][0]["v"])
