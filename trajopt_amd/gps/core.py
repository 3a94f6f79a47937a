"""Drop-in for `trajopt.gps.core.backward_pass`
(/root/reference/trajopt/gps/src/util.cpp:272-374, called at /root/reference/trajopt/gps/mbgps.py:157-160).

Positional signature and the 13-tuple of the pybind11 function; the work is done by
trajopt_gps_backward_pass_host() in libtrajopt_amd.so (HIP kernel on the current device).  The module's other
functions - forward_pass (util.cpp:195-269), augment_cost (:136-193), kl_divergence (:83-121), quad_expectation
(:123-134) - keep their signatures and return tuples the same way, over trajopt_gps_*_host().
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


def forward_pass(mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, dm_state, dm_act, nb_steps):
    """-> (mu_x, sigma_x, mu_u, sigma_u, mu_xu, sigma_xu), as util.cpp:259-268 (mbgps.py:140-145)."""
    require_gpu()
    n, m, T = int(dm_state), int(dm_act), int(nb_steps)
    mu_x0 = _in(mu_x0, (n,), "mu_x0"); sigma_x0 = _in(sigma_x0, (n, n), "sigma_x0")
    A = _in(A, (n, n, T), "A"); B = _in(B, (n, m, T), "B"); c = _in(c, (n, T), "c")
    sigma_dyn = _in(sigma_dyn, (n, n, T), "sigma_dyn")
    K = _in(K, (m, n, T), "K"); kff = _in(kff, (m, T), "kff"); sigma_ctl = _in(sigma_ctl, (m, m, T), "sigma_ctl")
    z = lambda *s: np.zeros(s, order="F")
    out = (z(n, T + 1), z(n, n, T + 1), z(m, T), z(m, m, T), z(n + m, T + 1), z(n + m, n + m, T + 1))
    check(lib().trajopt_gps_forward_pass_host(
        _ptr(mu_x0), _ptr(sigma_x0), _ptr(A), _ptr(B), _ptr(c), _ptr(sigma_dyn), _ptr(K), _ptr(kff), _ptr(sigma_ctl),
        n, m, T, *map(_ptr, out)))
    return out


def augment_cost(Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, dm_state, dm_act, nb_steps):
    """-> (agCxx, agcx, agCuu, agcu, agCxu, agc0), as util.cpp:184-192 (mbgps.py:166-170)."""
    require_gpu()
    n, m, T = int(dm_state), int(dm_act), int(nb_steps)
    Cxx = _in(Cxx, (n, n, T + 1), "Cxx"); cx = _in(cx, (n, T + 1), "cx")
    Cuu = _in(Cuu, (m, m, T + 1), "Cuu"); cu = _in(cu, (m, T + 1), "cu")
    Cxu = _in(Cxu, (n, m, T + 1), "Cxu"); c0 = _in(c0, (T + 1,), "c0")
    K = _in(K, (m, n, T), "K"); kff = _in(kff, (m, T), "kff"); sigma_ctl = _in(sigma_ctl, (m, m, T), "sigma_ctl")
    alpha = _in(alpha, (T,), "alpha")
    z = lambda *s: np.zeros(s, order="F")
    out = (z(n, n, T + 1), z(n, T + 1), z(m, m, T + 1), z(m, T + 1), z(n, m, T + 1), z(T + 1))
    status = C.c_int(0)
    check(lib().trajopt_gps_augment_cost_host(
        _ptr(Cxx), _ptr(cx), _ptr(Cuu), _ptr(cu), _ptr(Cxu), _ptr(c0), _ptr(K), _ptr(kff), _ptr(sigma_ctl), _ptr(alpha),
        n, m, T, *map(_ptr, out), C.byref(status)))
    if status.value:  # Armadillo's inv_sympd() throws std::runtime_error (util.cpp:164)
        raise RuntimeError("inv_sympd(): matrix is singular or not positive definite (sigma_ctl)")
    return out


def kl_divergence(p_K, p_kff, p_sigma_ctl, q_K, q_kff, q_sigma_ctl, mu_x, sigma_x, dm_state, dm_act, nb_steps):
    """-> (kl,), a 1-tuple as util.cpp:119 (mbgps.py:198-201 passes T+1 slices of mu_x / sigma_x; T are read)."""
    require_gpu()
    n, m, T = int(dm_state), int(dm_act), int(nb_steps)
    p_K = _in(p_K, (m, n, T), "p_K"); p_kff = _in(p_kff, (m, T), "p_kff")
    p_sigma_ctl = _in(p_sigma_ctl, (m, m, T), "p_sigma_ctl")
    q_K = _in(q_K, (m, n, T), "q_K"); q_kff = _in(q_kff, (m, T), "q_kff")
    q_sigma_ctl = _in(q_sigma_ctl, (m, m, T), "q_sigma_ctl")
    mu_x = np.asfortranarray(np.asarray(mu_x, dtype=np.float64))
    sigma_x = np.asfortranarray(np.asarray(sigma_x, dtype=np.float64))
    if mu_x.shape[:1] != (n,) or mu_x.ndim != 2 or mu_x.shape[1] < T:
        raise ValueError("mu_x has shape %s, expected (%d, >= %d)" % (mu_x.shape, n, T))
    if sigma_x.shape[:2] != (n, n) or sigma_x.ndim != 3 or sigma_x.shape[2] < T:
        raise ValueError("sigma_x has shape %s, expected (%d, %d, >= %d)" % (sigma_x.shape, n, n, T))
    kl = np.zeros(T)
    status = C.c_int(0)
    check(lib().trajopt_gps_kl_divergence_host(
        _ptr(p_K), _ptr(p_kff), _ptr(p_sigma_ctl), _ptr(q_K), _ptr(q_kff), _ptr(q_sigma_ctl), _ptr(mu_x),
        _ptr(sigma_x), n, m, T, _ptr(kl), C.byref(status)))
    if status.value:
        raise RuntimeError("inv_sympd(): matrix is singular or not positive definite (q_sigma_ctl)")
    return (kl,)


def quad_expectation(mu, sigma_s, Q, q, q0):
    """-> float, util.cpp:123-134."""
    require_gpu()
    mu = np.asfortranarray(np.asarray(mu, dtype=np.float64))
    n = int(mu.shape[0])
    mu = _in(mu, (n,), "mu"); sigma_s = _in(sigma_s, (n, n), "sigma_s"); Q = _in(Q, (n, n), "Q"); q = _in(q, (n,), "q")
    res = C.c_double(0.0)
    check(lib().trajopt_gps_quad_expectation_host(_ptr(mu), _ptr(sigma_s), _ptr(Q), _ptr(q), C.c_double(float(q0)), n,
                                                  C.byref(res)))
    return float(res.value)
