"""The reference's import path resolves to this library (/root/reference/trajopt/ilqr/ilqr.py:7)."""
import numpy as np
import pytest


def test_reference_import_path_resolves_without_a_gpu():
    from trajopt.ilqr import iLQR
    from trajopt.ilqr.core import backward_pass
    from trajopt.ilqr.objects import LinearControl, QuadraticStateValue  # noqa: F401
    from trajopt.riccati import Riccati
    from trajopt.gps.core import augment_cost, forward_pass, kl_divergence, quad_expectation  # noqa: F401
    from trajopt.gps.core import backward_pass as gps_backward_pass
    from trajopt.bspilqr.core import backward_pass as bsp_backward_pass
    import trajopt.envs
    import trajopt_amd.ilqr.core as C
    assert backward_pass is C.backward_pass
    assert iLQR.__module__.startswith("trajopt_amd") and Riccati.__module__.startswith("trajopt_amd")
    assert callable(gps_backward_pass) and callable(bsp_backward_pass) and callable(trajopt.envs.make)


@pytest.mark.gpu
def test_backward_pass_through_the_reference_import_path(oracle):
    from trajopt.ilqr.core import backward_pass
    rng = np.random.default_rng(5)
    n, m, T = 4, 1, 30
    A = np.asfortranarray(np.eye(n)[:, :, None] + 0.1 * rng.standard_normal((n, n, T)))
    B = np.asfortranarray(0.3 * rng.standard_normal((n, m, T)))
    Cxx = np.asfortranarray(np.tile(np.eye(n)[:, :, None], (1, 1, T + 1)))
    Cuu = np.asfortranarray(np.tile(0.1 * np.eye(m)[:, :, None], (1, 1, T + 1)))
    Cxu = np.zeros((n, m, T + 1), order="F")
    cx, cu = np.asfortranarray(rng.standard_normal((n, T + 1))), np.asfortranarray(rng.standard_normal((m, T + 1)))
    out = backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, 1.0, 1, n, m, T)
    ref = oracle.backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, 1.0, 1, n, m, T)
    ref = ref[0] if isinstance(ref[0], tuple) else ref
    assert len(out) == 11
    for k in (5, 6, 7, 8, 9):   # V, v, dV, K, kff
        got, exp = np.asarray(out[k]), np.asarray(ref[k])
        assert np.max(np.abs(got - exp)) <= 1e-9 * max(1.0, np.max(np.abs(exp))), k
