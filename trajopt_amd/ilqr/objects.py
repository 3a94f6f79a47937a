"""Containers of the reference's iLQR (/root/reference/trajopt/ilqr/objects.py), same names, same
NumPy shapes (time is the last axis).  The two `taylor_expansion` methods (objects.py:68-78,114-117) run
the fused HIP quadratisation kernel (trajopt_quadratize_batched) instead of autograd loops.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..layout import from_device, to_device


class QuadraticStateValue:  # objects.py:5-11
    def __init__(self, dm_state, nb_steps):
        self.dm_state, self.nb_steps = dm_state, nb_steps
        self.V = np.zeros((dm_state, dm_state, nb_steps))
        self.v = np.zeros((dm_state, nb_steps))


class QuadraticStateActionValue:  # objects.py:14-25
    def __init__(self, dm_state, dm_act, nb_steps):
        self.dm_state, self.dm_act, self.nb_steps = dm_state, dm_act, nb_steps
        self.Qxx = np.zeros((dm_state, dm_state, nb_steps))
        self.Quu = np.zeros((dm_act, dm_act, nb_steps))
        self.Qux = np.zeros((dm_act, dm_state, nb_steps))
        self.qx = np.zeros((dm_state, nb_steps))
        self.qu = np.zeros((dm_act, nb_steps))


class QuadraticCost:  # objects.py:28-49
    def __init__(self, dm_state, dm_act, nb_steps):
        self.dm_state, self.dm_act, self.nb_steps = dm_state, dm_act, nb_steps
        self.Cxx = np.zeros((dm_state, dm_state, nb_steps))
        self.cx = np.zeros((dm_state, nb_steps))
        self.Cuu = np.zeros((dm_act, dm_act, nb_steps))
        self.cu = np.zeros((dm_act, nb_steps))
        self.Cxu = np.zeros((dm_state, dm_act, nb_steps))

    @property
    def params(self):
        return self.Cxx, self.cx, self.Cuu, self.cu, self.Cxu

    @params.setter
    def params(self, values):
        self.Cxx, self.cx, self.Cuu, self.cu, self.Cxu = values


class LinearDynamics:  # objects.py:81-99
    def __init__(self, dm_state, dm_act, nb_steps):
        self.dm_state, self.dm_act, self.nb_steps = dm_state, dm_act, nb_steps
        self.A = np.zeros((dm_state, dm_state, nb_steps))
        self.B = np.zeros((dm_state, dm_act, nb_steps))

    @property
    def params(self):
        return self.A, self.B

    @params.setter
    def params(self, values):
        self.A, self.B = values

    def sample(self, x, u):
        pass


class LinearControl:  # objects.py:120-139
    def __init__(self, dm_state, dm_act, nb_steps):
        self.dm_state, self.dm_act, self.nb_steps = dm_state, dm_act, nb_steps
        self.K = np.zeros((dm_act, dm_state, nb_steps))
        self.kff = np.zeros((dm_act, nb_steps))

    @property
    def params(self):
        return self.K, self.kff

    @params.setter
    def params(self, values):
        self.K, self.kff = values

    def action(self, x, alpha, xref, uref, t):
        dx = x[..., t] - xref[:, t]
        return uref[:, t] + alpha * self.kff[..., t] + self.K[..., t] @ dx


def quadratize(env, nb_steps, xref, uref, weighting, convention=0):
    """One problem, host arrays in the reference layout -> dict(A, B, c, Cxx, cx, Cuu, cu, Cxu), computed by
    the HIP kernel.  convention 0 = iLQR (objects.py:68-78,114-117), 1 = Riccati (riccati/objects.py:54-65,102-108)."""
    import torch
    _lib.require_gpu()
    if not hasattr(env, "struct"):
        raise _lib.TrajoptError("the device quadratisation needs a built-in env model (trajopt_amd.envs); an arbitrary "
                                "Python env is expanded on the host by iLQR.run (trajopt_amd/ilqr/hostenv.py)")
    e = env.struct()
    n, m, T = e.dm_state, e.dm_act, int(nb_steps)
    dev = torch.device("cuda", torch.cuda.current_device())
    L = _lib.LAYOUT_PROBLEM_MINOR
    x = to_device(np.asarray(xref, dtype=np.float64).reshape(1, n, T + 1), L, dev)
    u = to_device(np.asarray(uref, dtype=np.float64).reshape(1, m, T), L, dev)
    w = torch.as_tensor(np.asarray(weighting, dtype=np.float64).reshape(T + 1), device=dev)
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)
    o = dict(A=z(T, n, n, 1), B=z(T, m, n, 1), c=z(T, n, 1), Cxx=z(T + 1, n, n, 1), cx=z(T + 1, n, 1),
             Cuu=z(T + 1, m, m, 1), cu=z(T + 1, m, 1), Cxu=z(T + 1, m, n, 1))
    _lib.check(_lib.lib().trajopt_quadratize_batched(
        C.byref(e), T, C.c_int64(1), L, int(convention), _lib.dptr(x), _lib.dptr(u), _lib.dptr(w), None,
        _lib.dptr(o["A"]), _lib.dptr(o["B"]), _lib.dptr(o["c"]), _lib.dptr(o["Cxx"]), _lib.dptr(o["cx"]),
        _lib.dptr(o["Cuu"]), _lib.dptr(o["cu"]), _lib.dptr(o["Cxu"]), _lib.stream_ptr()))
    return {k: np.ascontiguousarray(from_device(v, L)[0]) for k, v in o.items()}


class AnalyticalQuadraticCost(QuadraticCost):  # objects.py:52-78
    """`f` is the bound `env.cost`; nb_steps here is the reference's T + 1 (ilqr.py:85)."""

    def __init__(self, f, dm_state, dm_act, nb_steps):
        super().__init__(dm_state, dm_act, nb_steps)
        self.f = f
        self.env = getattr(f, "__self__", None)

    def evalf(self, x, u, u_last, a):
        return self.f(x, u, u_last, a)

    def taylor_expansion(self, x, u, a):
        q = quadratize(self.env, self.nb_steps - 1, x, u, a)
        self.Cxx, self.cx, self.Cuu, self.cu, self.Cxu = q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"]


class AnalyticalLinearDynamics(LinearDynamics):  # objects.py:102-117
    def __init__(self, f_dyn, dm_state, dm_act, nb_steps):
        super().__init__(dm_state, dm_act, nb_steps)
        self.f = f_dyn
        self.env = getattr(f_dyn, "__self__", None)

    def evalf(self, x, u):
        return self.f(x, u)

    def taylor_expansion(self, x, u):
        w = np.ones(self.nb_steps + 1)
        q = quadratize(self.env, self.nb_steps, x, u, w)
        self.A, self.B = q["A"], q["B"]
