"""Drop-in for the reference's native module `trajopt.ilqr.core`
(/root/reference/trajopt/ilqr/src/util.cpp:83-188, built by /root/reference/trajopt/ilqr/CMakeLists.txt).

`backward_pass` keeps the positional signature and the 11-tuple of the pybind11 function; the work is
done by trajopt_ilqr_backward_pass_host() in libtrajopt_amd.so (HIP kernels on the current device).
"""
import ctypes as C

import numpy as np

from .._lib import check, lib, require_gpu


def _in(a, shape, name):
    # pybind11's array_t<double, f_style | forcecast> accepts any array-like and converts (util.cpp:10)
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if a.shape != shape:
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, shape))
    return a


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, dm_state, dm_act, nb_steps):
    """-> (Qxx, Qux, Quu, qx, qu, V, v, dV, K, kff, diverge), as util.cpp:179-180."""
    out, _ = backward_pass_flag(Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, dm_state, dm_act, nb_steps)
    return out


def backward_pass_flag(Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, dm_state, dm_act, nb_steps):
    """Same, plus the unambiguous `failed` flag (the reference's diverge == 0 also means "failed at t = 0")."""
    require_gpu()
    n, m, T = int(dm_state), int(dm_act), int(nb_steps)
    Cxx = _in(Cxx, (n, n, T + 1), "Cxx"); cx = _in(cx, (n, T + 1), "cx")
    Cuu = _in(Cuu, (m, m, T + 1), "Cuu"); cu = _in(cu, (m, T + 1), "cu")
    Cxu = _in(Cxu, (n, m, T + 1), "Cxu"); A = _in(A, (n, n, T), "A"); B = _in(B, (n, m, T), "B")
    z = lambda *s: np.zeros(s, order="F")
    Qxx, Qux, Quu, qx, qu = z(n, n, T), z(m, n, T), z(m, m, T), z(n, T), z(m, T)
    V, v, dV, K, kff = z(n, n, T + 1), z(n, T + 1), z(2), z(m, n, T), z(m, T)
    diverge, failed = C.c_int(0), C.c_int(0)
    check(lib().trajopt_ilqr_backward_pass_host(
        _ptr(Cxx), _ptr(cx), _ptr(Cuu), _ptr(cu), _ptr(Cxu), _ptr(A), _ptr(B), C.c_double(float(lmbda)), int(reg),
        n, m, T, _ptr(Qxx), _ptr(Qux), _ptr(Quu), _ptr(qx), _ptr(qu), _ptr(V), _ptr(v), _ptr(dV), _ptr(K), _ptr(kff),
        C.byref(diverge), C.byref(failed)))
    return (Qxx, Qux, Quu, qx, qu, V, v, dV, K, kff, int(diverge.value)), bool(failed.value)
