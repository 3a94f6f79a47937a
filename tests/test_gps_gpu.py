"""GPU parity of the max-entropy (GPS) backward pass, SURVEY 8f.2, against the CPU oracle, through the C ABI.
Parity unpinned, like every row: the reference holds no vectors for gps/src/util.cpp and cannot be built here."""
import numpy as np
import pytest

from tests.util import assert_close

pytestmark = pytest.mark.gpu

OUT = ("Qxx", "Qux", "Quu", "qx", "qu", "q0", "V", "v", "v0", "K", "kff", "sigma_ctl")
ARGS = ("Cxx", "cx", "Cuu", "cu", "Cxu", "c0", "A", "B", "c", "sigma_dyn", "alpha")
SHAPES = [(2, 1), (3, 1), (4, 1), (6, 1), (2, 2), (4, 2)]


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


@pytest.mark.parametrize("n,m", [(4, 1), (4, 2)])
def test_gains_only_variant_writes_the_same_gains(oracle, n, m):
    from trajopt_amd import synthetic as S
    T, N = 50, 70
    stack = S.gps_inputs(n, m, T, N, seed=99)
    a, b = _run_gpu(stack, n, m, T, N, full=True), _run_gpu(stack, n, m, T, N, full=False)
    assert "Qxx" not in b
    for k in ("K", "kff", "sigma_ctl", "V", "v", "v0"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2)])
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


def test_unsupported_shapes_are_refused():
    from trajopt_amd._lib import lib
    assert lib().trajopt_gps_backward_supported(4, 1) == 1
    assert lib().trajopt_gps_backward_supported(12, 4) == 0
