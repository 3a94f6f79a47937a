"""GPU parity of the belief-space iLQR backward pass against the CPU oracle, through the C ABI.
Parity unpinned, like every row: the reference holds no vectors for bspilqr/src/util.cpp and cannot be built here."""
import numpy as np
import pytest

from tests.util import assert_close

pytestmark = pytest.mark.gpu

OUT = ("S", "s", "tau", "dS", "K", "kff")
ARGS = ("Q", "q", "R", "r", "P", "p", "F", "G", "T", "U", "V", "X", "Y", "Z")


def _run_gpu(stack, lmbda, reg, b, m, T, N, tau_mode=0):
    import torch
    from trajopt_amd import backward, layout as L
    d = {k: L.to_device(stack[k], 1) for k in ARGS}
    lam = torch.full((N,), float(lmbda), dtype=torch.float64, device="cuda")
    o = backward.bsp_backward_pass(*[d[k] for k in ARGS], lam, reg, b, m, T, N, 1, tau_mode=tau_mode)
    torch.cuda.synchronize()
    res = {k: L.from_device(o[k], 1) for k in OUT}
    res["diverge"] = o["diverge"].cpu().numpy()
    return res


@pytest.mark.parametrize("b,m", [(2, 1), (2, 2), (4, 1), (4, 2)])
@pytest.mark.parametrize("T,N", [(30, 67), (5, 130)])
@pytest.mark.parametrize("reg", [1, 2])
@pytest.mark.parametrize("tau_mode", [0, 1])
def test_bsp_backward_matches_oracle(oracle, b, m, T, N, reg, tau_mode):
    from trajopt_amd import synthetic as S
    stack = S.bsp_inputs(b, m, T, N, seed=31 + 10 * b + m, coupling=5e-2)
    res = _run_gpu(stack, 0.5, reg, b, m, T, N, tau_mode)
    for p in range(0, N, 2):
        q = S.problem(stack, p)
        ref, failed = oracle.bsp_backward_pass(*[q[k] for k in ARGS], 0.5, reg, b, m, T, tau_mode=tau_mode)
        assert not failed and res["diverge"][p] == -1
        for k, r in zip(OUT, ref[:6]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))


# shapes without an instantiated lane-per-problem kernel: the wave-per-problem kernel with run-time dimensions
# (gen_wave.hip; bspilqr/src/util.cpp:83 takes any dimensions)
@pytest.mark.parametrize("b,m", [(3, 1), (6, 2), (5, 3), (2, 4), (8, 4), (1, 1)])
@pytest.mark.parametrize("reg", [1, 2])
@pytest.mark.parametrize("tau_mode", [0, 1])
def test_bsp_backward_any_dimensions_match_oracle(oracle, b, m, reg, tau_mode):
    from trajopt_amd import synthetic as S
    T, N = 12, 5
    stack = S.bsp_inputs(b, m, T, N, seed=57 + 10 * b + m, coupling=2e-2)
    res = _run_gpu(stack, 0.5, reg, b, m, T, N, tau_mode)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.bsp_backward_pass(*[q[k] for k in ARGS], 0.5, reg, b, m, T, tau_mode=tau_mode)
        assert not failed and res["diverge"][p] == -1
        for k, r in zip(OUT, ref[:6]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))


@pytest.mark.parametrize("b,m", [(4, 2), (6, 3)])
def test_bsp_divergence_index_and_partial_outputs(oracle, b, m):
    from trajopt_amd import synthetic as S
    T, N, at = 20, 66, {0: 11, 7: 0, 65: 19}
    stack = {k: np.array(v) for k, v in S.bsp_inputs(b, m, T, N, seed=3).items()}
    for p, t in at.items():
        stack["R"][p, :, :, t] = -50.0 * np.eye(m)
    res = _run_gpu(stack, 0.5, 1, b, m, T, N)
    for p in range(N):
        q = S.problem(stack, p)
        ref, failed = oracle.bsp_backward_pass(*[q[k] for k in ARGS], 0.5, 1, b, m, T)
        assert res["diverge"][p] == (at[p] if p in at else -1) and bool(failed) == (p in at)
        for k, r in zip(OUT, ref[:6]):
            assert_close(res[k][p], r, "%s[p=%d]" % (k, p))


def test_bsp_host_entry_point_is_the_reference_signature(oracle):
    from trajopt_amd.bspilqr import core
    from trajopt_amd import synthetic as S
    from trajopt_amd._lib import lib
    b, m, T = 4, 2, 25
    q = S.problem(S.bsp_inputs(b, m, T, 1, seed=77), 0)
    pad = lambda a: np.concatenate([a, np.zeros(a.shape[:-1] + (1,))], axis=-1)   # the reference's containers: T + 1 slices
    args = [q[k] if k in ("Q", "q", "p") else pad(q[k]) for k in ARGS]
    out = core.backward_pass(*args, 0.5, 1, b, m, T)
    ref, failed = oracle.bsp_backward_pass(*[q[k] for k in ARGS], 0.5, 1, b, m, T)
    assert len(out) == 7 and out[6] == 0 and not failed
    for k, a, r in zip(OUT, out[:6], ref[:6]):
        assert a.flags.f_contiguous and a.shape == r.shape
        assert_close(a, r, k)
    q["R"][:, :, 9] = -50.0 * np.eye(m)
    out, flag = core.backward_pass_flag(*[q[k] for k in ARGS], 0.5, 1, b, m, T)
    assert out[6] == 9 and flag
    assert lib().trajopt_bspilqr_backward_supported(4, 2) == 1 and lib().trajopt_bspilqr_backward_supported(6, 2) == 1
    assert lib().trajopt_bspilqr_backward_supported(9, 2) == 0 and lib().trajopt_bspilqr_backward_supported(4, 9) == 0
