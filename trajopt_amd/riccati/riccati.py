"""Time-varying LQR solvers.

`Riccati`        - the reference's class (/root/reference/trajopt/riccati/riccati.py:8-161), one problem.
`BatchedRiccati` - the backward sweep (riccati.py:100-126) for N problems with caller-supplied
                   quadratic/affine models resident on the GPU (BASELINE config C5).
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import check, lib
from ..backward import riccati_backward_pass
from ..ilqr.ilqr import make_weighting
from .objects import AnalyticalLinearDynamics, AnalyticalQuadraticCost, LinearControl, QuadraticStateValue


def _in(a, shape, name):
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if a.shape != shape:
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, shape))
    return a


def backward_pass_host(Cxx, cx, Cuu, cu, Cxu, A, B, c, dm_state, dm_act, nb_steps):
    """riccati.py:100-126 on host arrays in the reference layout -> (K, kff, V, v); runs the HIP kernel.
    Raises numpy.linalg.LinAlgError where np.linalg.inv (riccati.py:118) would."""
    _lib.require_gpu()
    n, m, T = int(dm_state), int(dm_act), int(nb_steps)
    Cxx = _in(Cxx, (n, n, T + 1), "Cxx"); cx = _in(cx, (n, T + 1), "cx")
    Cuu = _in(Cuu, (m, m, T + 1), "Cuu"); cu = _in(cu, (m, T + 1), "cu")
    Cxu = _in(Cxu, (n, m, T + 1), "Cxu"); A = _in(A, (n, n, T), "A"); B = _in(B, (n, m, T), "B")
    c = _in(c, (n, T), "c")
    z = lambda *s: np.zeros(s, order="F")
    V, v, K, kff = z(n, n, T + 1), z(n, T + 1), z(m, n, T), z(m, T)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    status = C.c_int(0)
    check(lib().trajopt_riccati_backward_pass_host(p(Cxx), p(cx), p(Cuu), p(cu), p(Cxu), p(A), p(B), p(c), n, m, T,
                                                   p(V), p(v), p(K), p(kff), C.byref(status)))
    if status.value != 0:
        raise np.linalg.LinAlgError("Singular matrix (Quu at step %d)" % (status.value - 1))
    return K, kff, V, v


class Riccati:
    """Drop-in for trajopt.riccati.Riccati (riccati.py:8-161).  One deliberate deviation: V = Qxx + Qux^T K
    is the matrix product (riccati.py:123 writes an elementwise `*`, which only coincides for dm_act == 1 and
    raises a broadcast error otherwise)."""

    def __init__(self, env, nb_steps, init_state, activation=None):
        self.env = env
        self.env_dyn = self.env.unwrapped.dynamics
        self.env_noise = self.env.unwrapped.sigma
        self.env_cost = self.env.unwrapped.cost
        self.env_init = init_state
        self.ulim = self.env.action_space.high
        self.dm_state = self.env.observation_space.shape[0]
        self.dm_act = self.env.action_space.shape[0]
        self.nb_steps = nb_steps

        self.xref = np.zeros((self.dm_state, self.nb_steps + 1))
        self.xref[..., 0] = self.env_init[0]
        self.uref = np.zeros((self.dm_act, self.nb_steps))

        self.vfunc = QuadraticStateValue(self.dm_state, self.nb_steps + 1)
        self.dyn = AnalyticalLinearDynamics(self.env_dyn, self.dm_state, self.dm_act, self.nb_steps)
        self.ctl = LinearControl(self.dm_state, self.dm_act, self.nb_steps)
        self.weighting = make_weighting(self.nb_steps, activation)
        self.cost = AnalyticalQuadraticCost(self.env_cost, self.dm_state, self.dm_act, self.nb_steps + 1)

    def _episodes(self, ctl, nb_episodes, env):
        """`nb_episodes` closed-loop episodes of `ctl` on the noisy env, integrated by ONE kernel launch
        (trajopt_riccati_forward_pass_batched): the initial states and the per-step noise are drawn here, with the
        env's own generator in the order the reference's loops consume it (reset, then one draw per step, episode after
        episode: riccati.py:66-72,91-95), uploaded once and added on the device.
        The noise of an episode is drawn in one call with u.noise() taken once: that equals the reference's per-step
        draws while noise(x, u) is constant, which holds for every env of the reference (e.g. lqr_v0.py:68-69).
        -> state (n, T + 1, N), action (m, T, N), xnext (n, T, N), cost (T + 1, N)"""
        import torch
        _lib.require_gpu()
        u = env.unwrapped if hasattr(env, "unwrapped") else env
        if not hasattr(u, "struct"):
            raise _lib.TrajoptError(
                "Riccati.rollout / forward_pass integrate the episodes on the device and need an env with a device model "
                "(trajopt_amd.envs); %s has no struct()" % type(u).__name__)
        N, n, m, T = int(nb_episodes), self.dm_state, self.dm_act, self.nb_steps
        x0, noise = np.zeros((N, n)), np.zeros((N, T, n))
        for ep in range(N):
            x0[ep] = env.reset()
            noise[ep] = u.np_random.multivariate_normal(np.zeros(n), u.noise(), size=T)   # mean + these = env.step's draws
        dev = torch.device("cuda", torch.cuda.current_device())
        f64 = dict(dtype=torch.float64, device=dev)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), **f64)
        x0d, nzd = t(x0.T), t(noise.transpose(1, 2, 0))                       # (n, N), (T, n, N)
        Kd = t(np.asarray(ctl.K, dtype=np.float64).reshape(m, n, T).flatten(order="F"))
        kd = t(np.asarray(ctl.kff, dtype=np.float64).reshape(m, T).flatten(order="F"))
        wd = t(np.asarray(self.weighting, dtype=np.float64))
        st, ac = torch.empty((T + 1, n, N), **f64), torch.empty((T, m, N), **f64)
        xn, co = torch.empty((T, n, N), **f64), torch.empty((T + 1, N), **f64)
        e = u.struct()
        check(lib().trajopt_riccati_forward_pass_batched(
            C.byref(e), T, C.c_int64(N), _lib.dptr(x0d), _lib.dptr(Kd), _lib.dptr(kd), _lib.dptr(nzd), _lib.dptr(wd),
            _lib.dptr(st), _lib.dptr(ac), _lib.dptr(xn), _lib.dptr(co), _lib.stream_ptr()))
        st, ac, xn, co = (a.cpu().numpy() for a in (st, ac, xn, co))
        u.state = st[T, :, N - 1].copy()                                       # where the last episode left the env
        return st.transpose(1, 0, 2), ac.transpose(1, 0, 2), xn.transpose(1, 0, 2), co

    def rollout(self, nb_episodes, env=None):  # riccati.py:52-83
        state, action, xnext, cost = self._episodes(self.ctl, nb_episodes, self.env if env is None else env)
        c = cost.copy()
        c[:-1] = cost[:-1, -1:]   # riccati.py:71 assigns each step's cost to the whole row: the last episode's stays
        return {'x': np.ascontiguousarray(state[:, :-1]), 'u': np.ascontiguousarray(action),
                'xn': np.ascontiguousarray(xnext), 'c': c}

    def forward_pass(self, ctl):  # riccati.py:85-98: on the NOISY env (env.reset / env.step)
        state, action, _, cost = self._episodes(ctl, 1, self.env)
        return np.ascontiguousarray(state[..., 0]), np.ascontiguousarray(action[..., 0]), cost[:, 0].copy()

    def backward_pass(self):  # riccati.py:100-126
        lc = LinearControl(self.dm_state, self.dm_act, self.nb_steps)
        xvalue = QuadraticStateValue(self.dm_state, self.nb_steps + 1)
        lc.K, lc.kff, xvalue.V, xvalue.v = backward_pass_host(
            self.cost.Cxx, self.cost.cx, self.cost.Cuu, self.cost.cu, self.cost.Cxu, self.dyn.A, self.dyn.B,
            self.dyn.c, self.dm_state, self.dm_act, self.nb_steps)
        return lc, xvalue

    def run(self):  # riccati.py:148-161
        self.dyn.taylor_expansion(self.xref, self.uref)
        self.cost.taylor_expansion(self.xref, self.uref, self.weighting)
        self.ctl, self.vfunc = self.backward_pass()
        self.xref, self.uref, _cost = self.forward_pass(self.ctl)
        return np.sum(_cost)


class BatchedRiccati:
    """Backward sweep for N problems given device tensors (layout per trajopt_amd.layout) of
    Cxx, cx, Cuu, cu, Cxu (T+1 slices) and A, B, c (T slices)."""

    def __init__(self, dm_state, dm_act, nb_steps, nb_problems, layout=_lib.LAYOUT_PROBLEM_MAJOR):
        self.dm_state, self.dm_act, self.nb_steps, self.nb_problems = dm_state, dm_act, nb_steps, nb_problems
        self.layout = layout

    def backward_pass(self, Cxx, cx, Cuu, cu, Cxu, A, B, c):
        return riccati_backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, c, self.dm_state, self.dm_act, self.nb_steps,
                                     self.nb_problems, self.layout)
