"""Drop-in for `trajopt.gps.core.backward_pass`
(/root/reference/trajopt/gps/src/util.cpp:272-374, called at /root/reference/trajopt/gps/mbgps.py:157-160).

Positional signature and the 13-tuple of the pybind11 function; the work is done by
trajopt_gps_backward_pass_host() in libtrajopt_amd.so (HIP kernel on the current device).  The other functions
of that module (forward_pass, augment_cost, kl_divergence, quad_expectation) are not on this path yet.
"""
import ctypes as C

import numpy as np

from .._lib import check, lib, require_gpu
from ..ilqr.core import _in, _ptr


def backward_pass(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha, dm_state, dm_act, nb_steps):
    """-> (Qxx, Qux, Quu, qx, qu, q0, V, v, v0, K, kff, sigma_ctl, diverge), as util.cpp:355-371."""
    out, status = backward_pass_status(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha, dm_state, dm_act,
                                       nb_steps)
    if status == 2:  # Armadillo's inv() throws std::runtime_error on a singular Quu (util.cpp:338)
        raise RuntimeError("inv(): matrix is singular (Quu at step %d)" % out[-1])
    return out


def backward_pass_status(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha, dm_state, dm_act, nb_steps):
    """Same, plus status: 0 ok, 1 Quu positive definite at `diverge` (also when that index is 0), 2 singular Quu."""
    require_gpu()
    n, m, T = int(dm_state), int(dm_act), int(nb_steps)
    Cxx = _in(Cxx, (n, n, T + 1), "Cxx"); cx = _in(cx, (n, T + 1), "cx")
    Cuu = _in(Cuu, (m, m, T + 1), "Cuu"); cu = _in(cu, (m, T + 1), "cu")
    Cxu = _in(Cxu, (n, m, T + 1), "Cxu"); c0 = _in(c0, (T + 1,), "c0")
    A = _in(A, (n, n, T), "A"); B = _in(B, (n, m, T), "B"); c = _in(c, (n, T), "c")
    sigma_dyn = _in(sigma_dyn, (n, n, T), "sigma_dyn"); alpha = _in(alpha, (T,), "alpha")
    z = lambda *s: np.zeros(s, order="F")
    Qxx, Qux, Quu, qx, qu, q0 = z(n, n, T), z(m, n, T), z(m, m, T), z(n, T), z(m, T), z(T)
    V, v, v0, K, kff, sg = z(n, n, T + 1), z(n, T + 1), z(T + 1), z(m, n, T), z(m, T), z(m, m, T)
    diverge, status = C.c_int(0), C.c_int(0)
    check(lib().trajopt_gps_backward_pass_host(
        _ptr(Cxx), _ptr(cx), _ptr(Cuu), _ptr(cu), _ptr(Cxu), _ptr(c0), _ptr(A), _ptr(B), _ptr(c), _ptr(sigma_dyn),
        _ptr(alpha), n, m, T, _ptr(Qxx), _ptr(Qux), _ptr(Quu), _ptr(qx), _ptr(qu), _ptr(q0), _ptr(V), _ptr(v),
        _ptr(v0), _ptr(K), _ptr(kff), _ptr(sg), C.byref(diverge), C.byref(status)))
    return (Qxx, Qux, Quu, qx, qu, q0, V, v, v0, K, kff, sg, int(diverge.value)), int(status.value)
