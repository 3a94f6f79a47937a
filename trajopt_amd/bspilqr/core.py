"""Drop-in for `trajopt.bspilqr.core.backward_pass`
(/root/reference/trajopt/bspilqr/src/util.cpp:83-204, called at /root/reference/trajopt/bspilqr/bspilqr.py:90-104).

Same 19 positional arguments and the same 7-tuple (S, s, tau, dS, K, kff, diverge); the work is done by
trajopt_bspilqr_backward_pass_host() in libtrajopt_amd.so.  `tau_mode` (keyword, default 0) selects what the
reference's read of the not-yet-assigned tau.col(i) at util.cpp:158 means: 0 = tau.col(i+1), 1 = zeros.
"""
import ctypes as C

import numpy as np

from .._lib import check, lib, require_gpu
from ..ilqr.core import _ptr


def _arr(a, lead, slices, name):
    """F-ordered float64 with leading dims `lead` and at least `slices` trailing slices (the reference's cost
    containers hold nb_steps + 1 slices of everything, bspilqr/objects.py:93-100; only the needed prefix is read)."""
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if a.shape[:-1] != lead or a.shape[-1] < slices:
        raise ValueError("%s has shape %s, expected %s + (>= %d,)" % (name, a.shape, lead, slices))
    return a


def backward_pass(Q, q, R, r, P, p, F, G, T, U, V, X, Y, Z, lmbda, reg, dm_belief, dm_act, nb_steps, tau_mode=0):
    out, _ = backward_pass_flag(Q, q, R, r, P, p, F, G, T, U, V, X, Y, Z, lmbda, reg, dm_belief, dm_act, nb_steps,
                                tau_mode=tau_mode)
    return out


def backward_pass_flag(Q, q, R, r, P, p, F, G, T, U, V, X, Y, Z, lmbda, reg, dm_belief, dm_act, nb_steps, tau_mode=0):
    """Same, plus the unambiguous `failed` flag (the reference's diverge == 0 also means "failed at t = 0")."""
    require_gpu()
    b, m, n = int(dm_belief), int(dm_act), int(nb_steps)
    bb = b * b
    Q = _arr(Q, (b, b), n + 1, "Q"); q = _arr(q, (b,), n + 1, "q"); p = _arr(p, (bb,), n + 1, "p")
    R = _arr(R, (m, m), n, "R"); r = _arr(r, (m,), n, "r"); P = _arr(P, (b, m), n, "P")
    F = _arr(F, (b, b), n, "F"); G = _arr(G, (b, m), n, "G")
    T = _arr(T, (bb, b), n, "T"); U = _arr(U, (bb, bb), n, "U"); V = _arr(V, (bb, m), n, "V")
    X = _arr(X, (bb, b), n, "X"); Y = _arr(Y, (bb, bb), n, "Y"); Z = _arr(Z, (bb, m), n, "Z")
    z = lambda *s: np.zeros(s, order="F")
    S, s, tau, dS, K, kff = z(b, b, n + 1), z(b, n + 1), z(bb, n + 1), z(2), z(m, b, n), z(m, n)
    diverge, failed = C.c_int(0), C.c_int(0)
    check(lib().trajopt_bspilqr_backward_pass_host(
        *map(_ptr, (Q, q, R, r, P, p, F, G, T, U, V, X, Y, Z)), C.c_double(float(lmbda)), int(reg), int(tau_mode),
        b, m, n, *map(_ptr, (S, s, tau, dS, K, kff)), C.byref(diverge), C.byref(failed)))
    return (S, s, tau, dS, K, kff, int(diverge.value)), bool(failed.value)
