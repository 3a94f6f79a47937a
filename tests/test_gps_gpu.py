"""GPU parity of the max-entropy (GPS) backward pass, SURVEY 8f.2, against the CPU oracle, through the C ABI.
Parity unpinned, like every row: the reference holds no vectors for gps/src/util.cpp and cannot be built here."""
import numpy as np
import pytest

from tests.util import assert_close

pytestmark = pytest.mark.gpu

OUT = ("Qxx", "Qux", "Quu", "qx", "qu", "q0", "V", "v", "v0", "K", "kff", "sigma_ctl")
ARGS = ("Cxx", "cx", "Cuu", "cu", "Cxu", "c0", "A", "B", "c", "sigma_dyn", "alpha")
SHAPES = [(2, 1), (3, 1), (4, 1), (6, 1), (2, 2), (4, 2)]
FWD_OUT = ("mu_x", "sigma_x", "mu_u", "sigma_u", "mu_xu", "sigma_xu")
AG_OUT = ("Cxx", "cx", "Cuu", "cu", "Cxu", "c0")


def _run_gpu(stack, n, m, T, N, full=True):
    import torch
    from trajopt_amd import backward, layout as L
    d = {k: L.to_device(stack[k], 1) for k in ARGS}
    o = backward.gps_backward_pass(*[d[k] for k in ARGS], n, m, T, N, 1, full=full)
    torch.cuda.synchronize()
    res = {k: L.from_device(o[k], 1) for k in o if k not in ("diverge", "status")}
    res["diverge"], res["status"] = o["diverge"].cpu().numpy(), o["status"].cpu().numpy()
    return res


@pytest.mark.parametrize("n,m", SHAPES)
@pytest.mark.parametrize("T,N", [(40, 67), (100, 3), (7, 130)])
def test_gps_backward_matches_oracle(oracle, n, m, T, N):
    from trajopt_amd import synthetic as S
    stack = S.gps_inputs(n, m, T, N, seed=4242 + 10 * n + m)
    res = _run_gpu(stack, n, m, T, N)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.gps_backward_pass(*[q[k] for k in ARGS], n, m, T)
        assert not failed and res["diverge"][p] == -1 and res["status"][p] == 0
        for k, r in zip(OUT, ref[:12]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))


# shapes without an instantiated lane-per-problem kernel: the wave-per-problem kernel with run-time dimensions
# (gen_wave.hip; gps/src/util.cpp:272 takes any dimensions)
GENERIC = [(5, 3), (12, 4), (3, 4), (8, 1), (16, 8), (1, 1), (32, 16)]


@pytest.mark.parametrize("n,m", GENERIC)
@pytest.mark.parametrize("T,N", [(25, 9), (6, 3)])
def test_gps_backward_any_dimensions_match_oracle(oracle, n, m, T, N):
    from trajopt_amd import synthetic as S
    stack = S.gps_inputs(n, m, T, N, seed=777 + 10 * n + m)
    res = _run_gpu(stack, n, m, T, N)
    gains = _run_gpu(stack, n, m, T, N, full=False)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.gps_backward_pass(*[q[k] for k in ARGS], n, m, T)
        assert not failed and res["diverge"][p] == -1 and res["status"][p] == 0
        for k, r in zip(OUT, ref[:12]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))
    for k in ("K", "kff", "sigma_ctl", "V", "v", "v0"):
        assert np.array_equal(res[k], gains[k]), k


@pytest.mark.parametrize("n,m", [(4, 1), (4, 2)])
def test_gains_only_variant_writes_the_same_gains(oracle, n, m):
    from trajopt_amd import synthetic as S
    T, N = 50, 70
    stack = S.gps_inputs(n, m, T, N, seed=99)
    a, b = _run_gpu(stack, n, m, T, N, full=True), _run_gpu(stack, n, m, T, N, full=False)
    assert "Qxx" not in b
    for k in ("K", "kff", "sigma_ctl", "V", "v", "v0"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (5, 3), (12, 4)])
def test_positive_definite_quu_stops_the_sweep(oracle, n, m):
    """util.cpp:333-336: Quu = -(Cuu + B'VB)/alpha positive definite (a concave cost) -> diverge = t, the outputs of
    the steps already done stay, the rest are zero."""
    from trajopt_amd import synthetic as S
    T, N, at = 30, 66, {0: 17, 5: 0, 65: 29}
    stack = {k: np.array(v) for k, v in S.gps_inputs(n, m, T, N, seed=5).items()}
    for p, t in at.items():
        stack["Cuu"][p, :, :, t] = -50.0 * np.eye(m)
    res = _run_gpu(stack, n, m, T, N)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.gps_backward_pass(*[q[k] for k in ARGS], n, m, T)
        if p in at:
            assert failed and ref[12] == at[p]
            assert res["diverge"][p] == at[p] and res["status"][p] == 1
        else:
            assert not failed and res["diverge"][p] == -1 and res["status"][p] == 0
        for k, r in zip(OUT, ref[:12]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))


def test_host_entry_point_is_the_reference_signature(oracle):
    from trajopt_amd.gps import core
    from trajopt_amd import synthetic as S
    n, m, T = 4, 1, 60
    q = S.problem(S.gps_inputs(n, m, T, 1, seed=3), 0)
    out = core.backward_pass(*[q[k] for k in ARGS], n, m, T)
    ref, failed = oracle.gps_backward_pass(*[q[k] for k in ARGS], n, m, T)
    assert len(out) == 13 and out[12] == 0 and not failed
    for k, a, r in zip(OUT, out[:12], ref[:12]):
        assert a.flags.f_contiguous and a.shape == r.shape
        assert_close(a, r, k)
    q["Cuu"][:, :, 20] = -50.0
    out, status = core.backward_pass_status(*[q[k] for k in ARGS], n, m, T)
    assert out[12] == 20 and status == 1
    # an exactly singular Quu: B = 0 and Cuu = 0 at one step (Armadillo's inv() throws there, util.cpp:338)
    q["B"][:, :, 20] = 0.0
    q["Cuu"][:, :, 20] = 0.0
    with pytest.raises(RuntimeError):
        core.backward_pass(*[q[k] for k in ARGS], n, m, T)


def test_host_entry_points_take_any_dimensions(oracle):
    """The pybind11-shaped host entries on a shape without an instantiated kernel (dx = 5, du = 3): the reference's
    module takes any dimensions (gps/src/util.cpp:83-374)."""
    from trajopt_amd.gps import core
    from trajopt_amd import synthetic as S
    n, m, T = 5, 3, 20
    q = S.problem(S.gps_inputs(n, m, T, 1, seed=8), 0)
    out = core.backward_pass(*[q[k] for k in ARGS], n, m, T)
    ref, failed = oracle.gps_backward_pass(*[q[k] for k in ARGS], n, m, T)
    assert out[12] == 0 and not failed
    for k, a, r in zip(OUT, out[:12], ref[:12]):
        assert_close(a, r, k)
    K, kff, sg = ref[9], ref[10], ref[11]
    rng = np.random.default_rng(3)
    mu0, R = rng.standard_normal(n), rng.standard_normal((n, n))
    s0 = 1e-2 * (R.T @ R / n + np.eye(n))
    fw = core.forward_pass(mu0, s0, q["A"], q["B"], q["c"], q["sigma_dyn"], K, kff, sg, n, m, T)
    fw_ref = oracle.gps_forward_pass(mu0, s0, q["A"], q["B"], q["c"], q["sigma_dyn"], K, kff, sg, n, m, T)
    for k, a, r in zip(FWD_OUT, fw, fw_ref):
        assert_close(a, r, k)
    ag = core.augment_cost(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["c0"], K, kff, sg, q["alpha"], n, m, T)
    ag_ref = oracle.gps_augment_cost(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["c0"], K, kff, sg, q["alpha"], n, m, T)
    for k, a, r in zip(AG_OUT, ag, ag_ref):
        assert_close(a, r, "ag" + k)
    kl = core.kl_divergence(K + 0.1, kff - 0.1, 1.5 * sg, K, kff, sg, fw_ref[0], fw_ref[1], n, m, T)[0]
    assert_close(kl, oracle.gps_kl_divergence(K + 0.1, kff - 0.1, 1.5 * sg, K, kff, sg, fw_ref[0], fw_ref[1], n, m, T), "kl")
    qe = core.quad_expectation(mu0, s0, q["Cxx"][..., 0], q["cx"][:, 0], q["c0"][0])
    assert abs(qe - oracle.gps_quad_expectation(mu0, s0, q["Cxx"][..., 0], q["cx"][:, 0], q["c0"][0])) <= 1e-9 * max(1.0, abs(qe))


def test_unsupported_shapes_are_refused():
    from trajopt_amd._lib import lib
    assert lib().trajopt_gps_backward_supported(4, 1) == 1
    assert lib().trajopt_gps_backward_supported(12, 4) == 1     # run-time dimensions: dx <= 32, du <= 16
    assert lib().trajopt_gps_backward_supported(33, 4) == 0 and lib().trajopt_gps_backward_supported(4, 17) == 0


# ---- forward_pass, augment_cost, kl_divergence, quad_expectation (gps/src/util.cpp:83-269) ------------------------


def _policies(oracle, stack, n, m, T, N, seed):
    """Per problem: the oracle's controller for the problem, a perturbed second one, an initial state distribution."""
    from trajopt_amd import synthetic as S
    rng = np.random.Generator(np.random.PCG64(seed))
    K, kff, sg = np.zeros((N, m, n, T)), np.zeros((N, m, T)), np.zeros((N, m, m, T))
    for p in range(N):
        q = S.problem(stack, p)
        o, failed = oracle.gps_backward_pass(*[q[k] for k in ARGS], n, m, T)
        assert not failed
        K[p], kff[p], sg[p] = o[9], o[10], o[11]
    mu0 = rng.standard_normal((N, n))
    R = rng.standard_normal((N, n, n))
    s0 = 1e-2 * (np.einsum("pki,pkj->pij", R, R) / n + np.eye(n)[None])
    K2, k2, sg2 = K + 0.1 * rng.standard_normal(K.shape), kff + 0.1 * rng.standard_normal(kff.shape), 1.5 * sg
    return dict(K=K, kff=kff, sigma_ctl=sg, mu0=mu0, s0=s0, K2=K2, kff2=k2, sigma_ctl2=sg2)


@pytest.mark.parametrize("n,m", SHAPES)
@pytest.mark.parametrize("T,N", [(30, 67), (6, 130)])
def test_gps_forward_augment_kl_quadexp_match_oracle(oracle, n, m, T, N):
    _check_ops(oracle, n, m, T, N)


@pytest.mark.parametrize("n,m", [(5, 3), (12, 4), (3, 4), (8, 1), (16, 8), (1, 1)])
def test_gps_forward_augment_kl_quadexp_any_dimensions(oracle, n, m):
    """The same four functions on shapes without an instantiated kernel (gen_wave.hip: run-time dimensions)."""
    _check_ops(oracle, n, m, 12, 7)


def _check_ops(oracle, n, m, T, N):
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    stack = S.gps_inputs(n, m, T, N, seed=777 + 10 * n + m)
    pol = _policies(oracle, stack, n, m, T, N, seed=n + m)
    d = {k: L.to_device(v, 1) for k, v in {**stack, **pol}.items()}
    fw = backward.gps_forward_pass(d["mu0"], d["s0"], d["A"], d["B"], d["c"], d["sigma_dyn"], d["K"], d["kff"],
                                   d["sigma_ctl"], n, m, T, N)
    ag = backward.gps_augment_cost(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["c0"], d["K"], d["kff"],
                                   d["sigma_ctl"], d["alpha"], n, m, T, N)
    kl, kl_status = backward.gps_kl_divergence(d["K2"], d["kff2"], d["sigma_ctl2"], d["K"], d["kff"], d["sigma_ctl"],
                                               fw["mu_x"], fw["sigma_x"], n, m, T, N)   # T + 1 slices handed over
    qe = backward.gps_quad_expectation(d["mu0"], d["s0"], d["Cxx"][0].contiguous(), d["cx"][0].contiguous(),
                                       d["c0"][0].contiguous(), n, N)
    torch.cuda.synchronize()
    assert int(ag["status"].abs().sum()) == 0 and int(kl_status.abs().sum()) == 0
    fw_h = {k: L.from_device(fw[k], 1) for k in FWD_OUT}
    ag_h = {k: L.from_device(ag[k], 1) for k in AG_OUT}
    kl_h, qe_h = L.from_device(kl, 1), qe.cpu().numpy()
    for p in range(0, N, 3):
        q = S.problem(stack, p)
        ref = oracle.gps_forward_pass(pol["mu0"][p], pol["s0"][p], q["A"], q["B"], q["c"], q["sigma_dyn"], pol["K"][p],
                                      pol["kff"][p], pol["sigma_ctl"][p], n, m, T)
        for k, r in zip(FWD_OUT, ref):
            assert_close(fw_h[k][p], r, "%s[p=%d]" % (k, p))
        ref_ag = oracle.gps_augment_cost(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["c0"], pol["K"][p],
                                         pol["kff"][p], pol["sigma_ctl"][p], q["alpha"], n, m, T)
        for k, r in zip(AG_OUT, ref_ag):
            assert_close(ag_h[k][p], r, "ag%s[p=%d]" % (k, p))
        ref_kl = oracle.gps_kl_divergence(pol["K2"][p], pol["kff2"][p], pol["sigma_ctl2"][p], pol["K"][p],
                                          pol["kff"][p], pol["sigma_ctl"][p], ref[0], ref[1], n, m, T)
        assert_close(kl_h[p], ref_kl, "kl[p=%d]" % p)
        ref_qe = oracle.gps_quad_expectation(pol["mu0"][p], pol["s0"][p], q["Cxx"][..., 0], q["cx"][:, 0], q["c0"][0])
        assert abs(qe_h[p] - ref_qe) <= 1e-9 * max(1.0, abs(ref_qe))


def test_gps_dual_evaluation_chain_matches_oracle(oracle):
    """One MBGPS dual evaluation (gps/mbgps.py:173-195) with every array staying on the device:
    augment_cost -> backward_pass -> forward_pass -> quad_expectation -> kl_divergence."""
    import torch
    from trajopt_amd import backward, layout as L, synthetic as S
    n, m, T, N = 4, 1, 40, 70
    stack = S.gps_inputs(n, m, T, N, seed=2024)
    pol = _policies(oracle, stack, n, m, T, N, seed=5)   # the "last" controller self.ctl
    d = {k: L.to_device(v, 1) for k, v in {**stack, **pol}.items()}
    ag = backward.gps_augment_cost(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["c0"], d["K2"], d["kff2"],
                                   d["sigma_ctl2"], d["alpha"], n, m, T, N)
    bw = backward.gps_backward_pass(ag["Cxx"], ag["cx"], ag["Cuu"], ag["cu"], ag["Cxu"], ag["c0"], d["A"], d["B"],
                                    d["c"], d["sigma_dyn"], d["alpha"], n, m, T, N, 1, full=False)
    fw = backward.gps_forward_pass(d["mu0"], d["s0"], d["A"], d["B"], d["c"], d["sigma_dyn"], bw["K"], bw["kff"],
                                   bw["sigma_ctl"], n, m, T, N)
    dual = backward.gps_quad_expectation(d["mu0"], d["s0"], bw["V"][0].contiguous(), bw["v"][0].contiguous(),
                                         bw["v0"][0].contiguous(), n, N)
    kl, st = backward.gps_kl_divergence(bw["K"], bw["kff"], bw["sigma_ctl"], d["K2"], d["kff2"], d["sigma_ctl2"],
                                        fw["mu_x"], fw["sigma_x"], n, m, T, N)
    torch.cuda.synchronize()
    assert int(bw["diverge"].ne(-1).sum()) == 0 and int(st.abs().sum()) == 0
    dual_h, kl_h = dual.cpu().numpy(), L.from_device(kl, 1)
    for p in range(0, N, 7):
        q = S.problem(stack, p)
        rag = oracle.gps_augment_cost(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["c0"], pol["K2"][p],
                                      pol["kff2"][p], pol["sigma_ctl2"][p], q["alpha"], n, m, T)
        rb, failed = oracle.gps_backward_pass(*rag, q["A"], q["B"], q["c"], q["sigma_dyn"], q["alpha"], n, m, T)
        assert not failed
        rf = oracle.gps_forward_pass(pol["mu0"][p], pol["s0"][p], q["A"], q["B"], q["c"], q["sigma_dyn"], rb[9], rb[10],
                                     rb[11], n, m, T)
        rd = oracle.gps_quad_expectation(pol["mu0"][p], pol["s0"][p], rb[6][..., 0], rb[7][:, 0], rb[8][0])
        rk = oracle.gps_kl_divergence(rb[9], rb[10], rb[11], pol["K2"][p], pol["kff2"][p], pol["sigma_ctl2"][p], rf[0],
                                      rf[1], n, m, T)
        assert abs(dual_h[p] - rd) <= 1e-9 * max(1.0, abs(rd)), p
        assert_close(kl_h[p], rk, "kl[p=%d]" % p, tol=1e-8)   # differences of nearly equal gains: conditioning


def test_gps_core_module_signatures(oracle):
    from trajopt_amd.gps import core
    from trajopt_amd import synthetic as S
    n, m, T = 4, 2, 25
    stack = S.gps_inputs(n, m, T, 1, seed=8)
    pol = _policies(oracle, stack, n, m, T, 1, seed=1)
    q = S.problem(stack, 0)
    K, kff, sg = pol["K"][0], pol["kff"][0], pol["sigma_ctl"][0]
    fw = core.forward_pass(pol["mu0"][0], pol["s0"][0], q["A"], q["B"], q["c"], q["sigma_dyn"], K, kff, sg, n, m, T)
    ref = oracle.gps_forward_pass(pol["mu0"][0], pol["s0"][0], q["A"], q["B"], q["c"], q["sigma_dyn"], K, kff, sg, n, m, T)
    assert len(fw) == 6
    for k, a, r in zip(FWD_OUT, fw, ref):
        assert a.flags.f_contiguous and a.shape == r.shape
        assert_close(a, r, k)
    ag = core.augment_cost(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["c0"], K, kff, sg, q["alpha"], n, m, T)
    rag = oracle.gps_augment_cost(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["c0"], K, kff, sg, q["alpha"], n, m, T)
    assert len(ag) == 6
    for k, a, r in zip(AG_OUT, ag, rag):
        assert_close(a, r, "ag" + k)
    kl = core.kl_divergence(pol["K2"][0], pol["kff2"][0], pol["sigma_ctl2"][0], K, kff, sg, fw[0], fw[1], n, m, T)
    assert isinstance(kl, tuple) and len(kl) == 1 and kl[0].shape == (T,)
    assert_close(kl[0], oracle.gps_kl_divergence(pol["K2"][0], pol["kff2"][0], pol["sigma_ctl2"][0], K, kff, sg, ref[0],
                                                 ref[1], n, m, T), "kl")
    val = core.quad_expectation(pol["mu0"][0], pol["s0"][0], q["Cxx"][..., 0], q["cx"][:, 0], q["c0"][0])
    rv = oracle.gps_quad_expectation(pol["mu0"][0], pol["s0"][0], q["Cxx"][..., 0], q["cx"][:, 0], q["c0"][0])
    assert isinstance(val, float) and abs(val - rv) <= 1e-9 * max(1.0, abs(rv))
    bad = sg.copy(); bad[..., 4] = -np.eye(m)
    with pytest.raises(RuntimeError):
        core.augment_cost(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["c0"], K, kff, bad, q["alpha"], n, m, T)
    with pytest.raises(RuntimeError):
        core.kl_divergence(K, kff, sg, K, kff, bad, fw[0], fw[1], n, m, T)
