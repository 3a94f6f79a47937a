"""ORACLE - TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/riccati_oracle.hpp).

ctypes bindings of oracle/liboracle.so (built by `make -C oracle`, or __graft_entry__.build()).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (trajopt_amd/) never does.

Arrays follow the reference layout: per problem, column-major with time slowest - in NumPy terms
Fortran-ordered arrays of shape (rows, cols, T) as produced by the reference's pybind11 casters
(/root/reference/trajopt/ilqr/src/util.cpp:10-24).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_STATE, MAX_ACT = 12, 4
ENV_PENDULUM, ENV_CARTPOLE, ENV_DOUBLE_CARTPOLE, ENV_LQR, ENV_QUADROTOR, ENV_DOUBLE_PENDULUM, ENV_ROBOT, ENV_QUAD_PENDULUM = 1, 2, 3, 4, 5, 6, 7, 8


class EnvStruct(C.Structure):
    """Mirror of include/trajopt_env.h::trajopt_env_t."""
    _fields_ = [("model", C.c_int32), ("dm_state", C.c_int32), ("dm_act", C.c_int32),
                ("slew_rate", C.c_int32), ("periodic", C.c_int32), ("cost_times_dt", C.c_int32),
                ("nb_features", C.c_int32), ("features", C.c_int32 * MAX_STATE), ("reserved0", C.c_int32),
                ("dt", C.c_double),
                ("umax", C.c_double * MAX_ACT), ("xmax", C.c_double * MAX_STATE),
                ("goal", C.c_double * MAX_STATE), ("gw", C.c_double * MAX_STATE),
                ("uw", C.c_double * MAX_ACT),
                ("lin_A", C.c_double * (MAX_STATE * MAX_STATE)),
                ("lin_B", C.c_double * (MAX_STATE * MAX_ACT)),
                ("lin_c", C.c_double * MAX_STATE),
                ("phys", C.c_double * 16)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in
            ("oracle.cpp", "riccati_oracle.hpp", "envs_oracle.hpp", "ilqr_oracle.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.oracle_env_cost.restype = C.c_double
        _LIB.oracle_bench_backward_pass.restype = C.c_double
        _LIB.oracle_bench_ilqr.restype = C.c_double
    return _LIB


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


# ---- env parameter sets, constants cited from the reference -----------------------------------
# "...WithCartesianCost" variants: (features as (kind, state) with kind 0 = id, 1 = cos, 2 = sin; goal; weights)
#   pendulum.py:152-162, cartpole.py:150-166, double_cartpole.py:210-230 (which takes cos / sin of x[0], x[1] and
#   drops x[5] - kept as written), double_pendulum.py:178-194
CARTESIAN = {
    "pendulum": ([(1, 0), (2, 0), (0, 1)], [1., 0., 0.], [1e4, 1e4, 1e0]),
    "cartpole": ([(0, 0), (1, 1), (2, 1), (0, 2), (0, 3)], [0., 1., 0., 0., 0.], [1e1, 1e4, 1e4, 1e0, 1e0]),
    "double_cartpole": ([(0, 0), (1, 0), (2, 0), (1, 1), (2, 1), (0, 2), (0, 3), (0, 4)],
                        [0., 1., 0., 1., 0., 0., 0., 0.], [1e1, 1e4, 1e4, 1e4, 1e4, 1e0, 1e0, 1e0]),
    "double_pendulum": ([(1, 0), (2, 0), (1, 1), (2, 1), (0, 2), (0, 3)], [1., 0., 1., 0., 0., 0.],
                        [1e4, 1e4, 1e4, 1e4, 1e0, 1e0]),
    "quad_pendulum": ([(1, 0), (2, 0), (1, 1), (2, 1), (1, 2), (2, 2), (1, 3), (2, 3), (0, 4), (0, 5), (0, 6), (0, 7)],
                      [1., 0.] * 4 + [0.] * 4, [1e4] * 8 + [1e0] * 4),      # quad_pendulum.py:281-303
}


def make_env(name, dt=None, uw=None, slew_rate=False, periodic=False, cartesian=False, **kw):
    e = EnvStruct()
    inf = float("inf")

    def fill(field, vals):
        for i, v in enumerate(vals):
            field[i] = float(v)

    if name == "pendulum":  # envs/pendulum/pendulum.py:17-43,61
        e.model, e.dm_state, e.dm_act = ENV_PENDULUM, 2, 1
        e.dt = 0.01 if dt is None else dt
        fill(e.umax, [10.]); fill(e.xmax, [inf, inf]); fill(e.goal, [0., 0.]); fill(e.gw, [1e0, 1e-1])
        fill(e.uw, [1e-3] if uw is None else uw)
        fill(e.phys, [9.81, 1., 1., 1e-3])
        e.periodic = 0b01 if periodic else 0
    elif name == "cartpole":  # envs/cartpole/cartpole.py:17-47,67-72
        e.model, e.dm_state, e.dm_act = ENV_CARTPOLE, 4, 1
        e.dt = 0.01 if dt is None else dt
        fill(e.umax, [10.]); fill(e.xmax, [10., inf, inf, inf]); fill(e.goal, [0.] * 4)
        fill(e.gw, [1e1, 1e4, 1e0, 1e0]); fill(e.uw, [1e-5] if uw is None else uw)
        fill(e.phys, [9.81, 0.37, 0.127, 0.3365, 0.005])
        e.periodic = 0b0010 if periodic else 0
    elif name == "double_cartpole":  # envs/double_cartpole/double_cartpole.py:17-46,71-81
        e.model, e.dm_state, e.dm_act = ENV_DOUBLE_CARTPOLE, 6, 1
        e.dt = 0.01 if dt is None else dt
        fill(e.umax, [5.]); fill(e.xmax, [10.] + [inf] * 5); fill(e.goal, [0.] * 6)
        fill(e.gw, [1e1, 1e4, 1e4, 1e0, 1e0, 1e0]); fill(e.uw, [1e-5] if uw is None else uw)
        fill(e.phys, [9.81, 0.37, 0.127, 0.127, 0.3365, 0.3365])
        e.periodic = 0b000110 if periodic else 0
    elif name == "lqr":  # envs/lqr/lqr_v0.py:10-35
        A = np.asarray(kw.get("A", [[1.1, 0.], [1.0, 1.0]]), dtype=float)
        B = np.asarray(kw.get("B", [[1.], [0.]]), dtype=float)
        g = np.asarray(kw.get("goal", [10., 10.]), dtype=float)
        n, m = B.shape
        e.model, e.dm_state, e.dm_act = ENV_LQR, n, m
        e.dt = 0.1 if dt is None else dt
        e.cost_times_dt = 1
        fill(e.umax, [inf] * m); fill(e.xmax, [inf] * n); fill(e.goal, g)
        fill(e.gw, kw.get("gw", [1e1] * n)); fill(e.uw, [1e0] * m if uw is None else uw)
        fill(e.lin_A, A.flatten(order="F")); fill(e.lin_B, B.flatten(order="F"))
        fill(e.lin_c, kw.get("c", -A @ g))
    elif name == "double_pendulum":  # envs/double_pendulum/double_pendulum.py:17-46,66-70
        e.model, e.dm_state, e.dm_act = ENV_DOUBLE_PENDULUM, 4, 2
        e.dt = 0.01 if dt is None else dt
        fill(e.umax, [10., 10.]); fill(e.xmax, [inf] * 4); fill(e.goal, [0.] * 4)
        fill(e.gw, [1e4, 1e4, 1e0, 1e0]); fill(e.uw, [1e-5] * 2 if uw is None else uw)
        fill(e.phys, [9.81, 1., 1., 1., 1., 0.025, 0.025])
        e.periodic = 0b0001 if periodic else 0
    elif name == "quad_pendulum":  # envs/quad_pendulum/quad_pendulum.py:16-47,68-72
        e.model, e.dm_state, e.dm_act = ENV_QUAD_PENDULUM, 8, 4
        e.dt = 0.01 if dt is None else dt
        fill(e.umax, [25.] * 4); fill(e.xmax, [inf] * 8); fill(e.goal, [0.] * 8)
        fill(e.gw, [1e4] * 4 + [1e0] * 4); fill(e.uw, [1e-5] * 4 if uw is None else uw)
        fill(e.phys, [9.81] + [1.] * 4 + [1.] * 4 + [0.025] * 4)
        e.periodic = 0b00001111 if periodic else 0
    elif name == "robot":  # envs/robot/robot.py:10-36
        e.model, e.dm_state, e.dm_act = ENV_ROBOT, 4, 2
        e.dt = 0.025 if dt is None else dt
        e.cost_times_dt = 1
        fill(e.umax, [inf] * 2); fill(e.xmax, [inf] * 4); fill(e.goal, [0.] * 4)
        fill(e.gw, [1e1, 1e1, 1., 1.]); fill(e.uw, [1e-3] * 2 if uw is None else uw)
        fill(e.phys, [0.1])
    elif name in ("lqr_v1", "lqr_v2"):  # envs/lqr/lqr_v1.py:10-37, lqr_v2.py:10-36
        e.model, e.dm_state, e.dm_act = ENV_LQR, 2, 1
        e.dt = 0.01 if dt is None else dt
        e.cost_times_dt = 1
        if name == "lqr_v1":
            mm, k, d = 1., 1e-2, 1e-1
            A = np.array([[0., 1.], [-k / mm, -d / mm]]) * 0.01 + np.eye(2)   # built with the constructor's dt
            B = np.array([[0.], [1. / mm]]) * 0.01
            gw = [1e2, 1e0]
        else:
            A, B, gw = np.array([[1.0, 0.], [0.1, 1.1]]), np.array([[0.05], [0.]]), [1e1, 1e0]
        fill(e.umax, [inf]); fill(e.xmax, [inf] * 2); fill(e.goal, [1., 0.])
        fill(e.gw, gw); fill(e.uw, [1e-3] if uw is None else uw)
        fill(e.lin_A, A.flatten(order="F")); fill(e.lin_B, B.flatten(order="F")); fill(e.lin_c, [0., 0.])
    elif name == "quadrotor":  # build-defined, not in the reference
        e.model, e.dm_state, e.dm_act = ENV_QUADROTOR, 12, 4
        e.dt = 0.01 if dt is None else dt
        fill(e.umax, [5.] * 4); fill(e.xmax, [inf] * 12)
        fill(e.goal, [1., 1., 1.] + [0.] * 9)
        fill(e.gw, [1e1] * 3 + [1e0] * 3 + [1e0] * 3 + [1e-1] * 3)
        fill(e.uw, [1e-3] * 4 if uw is None else uw)
        fill(e.phys, [9.81, 0.5, 0.25, 3.9e-3, 3.9e-3, 7.5e-3, 1.6e-2])
    else:
        raise KeyError(name)
    if cartesian:
        feats, g, gw = CARTESIAN[name]
        e.nb_features = len(feats)
        for f, (kind, i) in enumerate(feats):
            e.features[f] = (kind << 8) | i
        for k in range(MAX_STATE):
            e.goal[k] = e.gw[k] = 0.0
        fill(e.goal, g); fill(e.gw, gw)
    e.slew_rate = int(bool(slew_rate))
    return e


# ---- A1 -----------------------------------------------------------------------------------------
def backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, dm_state, dm_act, nb_steps, long_double=False):
    """util.cpp:83-182.  Returns the reference's 11-tuple plus the unambiguous `failed` flag."""
    n, m, T = dm_state, dm_act, nb_steps
    Cxx, cx, Cuu, cu, Cxu, A, B = map(_f, (Cxx, cx, Cuu, cu, Cxu, A, B))
    assert Cxx.shape == (n, n, T + 1) and cx.shape == (n, T + 1) and Cuu.shape == (m, m, T + 1)
    assert cu.shape == (m, T + 1) and Cxu.shape == (n, m, T + 1) and A.shape == (n, n, T) and B.shape == (n, m, T)
    z = lambda *s: np.zeros(s, order="F")
    Qxx, Qux, Quu, qx, qu = z(n, n, T), z(m, n, T), z(m, m, T), z(n, T), z(m, T)
    V, v, dV, K, kff = z(n, n, T + 1), z(n, T + 1), z(2), z(m, n, T), z(m, T)
    failed = C.c_int(0)
    fn = lib().oracle_backward_pass_ld if long_double else lib().oracle_backward_pass
    div = fn(_p(Cxx), _p(cx), _p(Cuu), _p(cu), _p(Cxu), _p(A), _p(B), C.c_double(lmbda), int(reg), n, m, T,
             _p(Qxx), _p(Qux), _p(Quu), _p(qx), _p(qu), _p(V), _p(v), _p(dV), _p(K), _p(kff), C.byref(failed))
    return (Qxx, Qux, Quu, qx, qu, V, v, dV, K, kff, int(div)), bool(failed.value)


# ---- A9 -----------------------------------------------------------------------------------------
def riccati_backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, c, dm_state, dm_act, nb_steps, long_double=False,
                          literal_elementwise=False):
    """riccati.py:100-126.  Returns (K, kff, V, v)."""
    n, m, T = dm_state, dm_act, nb_steps
    Cxx, cx, Cuu, cu, Cxu, A, B, c = map(_f, (Cxx, cx, Cuu, cu, Cxu, A, B, c))
    z = lambda *s: np.zeros(s, order="F")
    V, v, K, kff = z(n, n, T + 1), z(n, T + 1), z(m, n, T), z(m, T)
    fn = lib().oracle_riccati_backward_pass_ld if long_double else lib().oracle_riccati_backward_pass
    rc = fn(_p(Cxx), _p(cx), _p(Cuu), _p(cu), _p(Cxu), _p(A), _p(B), _p(c), n, m, T, int(literal_elementwise),
            _p(V), _p(v), _p(K), _p(kff))
    if rc != 0:
        raise np.linalg.LinAlgError("singular Quu at step %d" % (rc - 1) if rc > 0 else "bad arguments")
    return K, kff, V, v


# ---- 8f.2 ---------------------------------------------------------------------------------------
GPS_OUT = ("Qxx", "Qux", "Quu", "qx", "qu", "q0", "V", "v", "v0", "K", "kff", "sigma_ctl")


def gps_backward_pass(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha, dm_state, dm_act, nb_steps,
                      long_double=False):
    """gps/src/util.cpp:272-374.  Returns (the reference's 13-tuple, failed)."""
    n, m, T = dm_state, dm_act, nb_steps
    Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha = map(_f, (Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn,
                                                                      alpha))
    assert c0.shape == (T + 1,) and c.shape == (n, T) and sigma_dyn.shape == (n, n, T) and alpha.shape == (T,)
    z = lambda *s: np.zeros(s, order="F")
    Qxx, Qux, Quu, qx, qu, q0 = z(n, n, T), z(m, n, T), z(m, m, T), z(n, T), z(m, T), z(T)
    V, v, v0, K, kff, sg = z(n, n, T + 1), z(n, T + 1), z(T + 1), z(m, n, T), z(m, T), z(m, m, T)
    failed = C.c_int(0)
    rc = lib().oracle_gps_backward_pass(_p(Cxx), _p(cx), _p(Cuu), _p(cu), _p(Cxu), _p(c0), _p(A), _p(B), _p(c),
                                        _p(sigma_dyn), _p(alpha), n, m, T, int(long_double), _p(Qxx), _p(Qux),
                                        _p(Quu), _p(qx), _p(qu), _p(q0), _p(V), _p(v), _p(v0), _p(K), _p(kff), _p(sg),
                                        C.byref(failed))
    if rc < 0:
        raise np.linalg.LinAlgError("singular Quu at step %d" % (-rc - 1))
    return (Qxx, Qux, Quu, qx, qu, q0, V, v, v0, K, kff, sg, int(rc)), bool(failed.value)


BSP_ARGS = ("Q", "q", "R", "r", "P", "p", "F", "G", "T", "U", "V", "X", "Y", "Z")


def bsp_backward_pass(Q, q, R, r, P, p, F, G, T_, U, V, X, Y, Z, lmbda, reg, dm_belief, dm_act, nb_steps, tau_mode=0,
                      long_double=False):
    """bspilqr/src/util.cpp:83-204.  Returns ((S, s, tau, dS, K, kff, diverge), failed)."""
    b, m, T = dm_belief, dm_act, nb_steps
    a = list(map(_f, (Q, q, R, r, P, p, F, G, T_, U, V, X, Y, Z)))
    z = lambda *s: np.zeros(s, order="F")
    S, s, tau, dS, K, kff = z(b, b, T + 1), z(b, T + 1), z(b * b, T + 1), z(2), z(m, b, T), z(m, T)
    failed = C.c_int(0)
    div = lib().oracle_bsp_backward_pass(*map(_p, a), C.c_double(float(lmbda)), int(reg), int(tau_mode), b, m, T,
                                         int(long_double), _p(S), _p(s), _p(tau), _p(dS), _p(K), _p(kff),
                                         C.byref(failed))
    return (S, s, tau, dS, K, kff, int(div)), int(failed.value)


def gps_kl_divergence(p_K, p_kff, p_sigma_ctl, q_K, q_kff, q_sigma_ctl, mu_x, sigma_x, dm_state, dm_act, nb_steps,
                      long_double=False):
    """gps/src/util.cpp:83-121 -> kl (T,)."""
    n, m, T = dm_state, dm_act, nb_steps
    a = list(map(_f, (p_K, p_kff, p_sigma_ctl, q_K, q_kff, q_sigma_ctl, mu_x, sigma_x)))
    kl = np.zeros(T)
    rc = lib().oracle_gps_kl_divergence(*map(_p, a), n, m, T, int(long_double), _p(kl))
    if rc:
        raise np.linalg.LinAlgError("inv_sympd(): q_sigma_ctl not positive definite at step %d" % (rc - 1))
    return kl


def gps_quad_expectation(mu, sigma_s, Q, q, q0):
    """gps/src/util.cpp:123-134 -> float."""
    mu, sigma_s, Q, q = map(_f, (mu, sigma_s, Q, q))
    fn = lib().oracle_gps_quad_expectation
    fn.restype = C.c_double
    return float(fn(_p(mu), _p(sigma_s), _p(Q), _p(q), C.c_double(float(q0)), int(mu.shape[0])))


def gps_augment_cost(Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, dm_state, dm_act, nb_steps,
                     long_double=False):
    """gps/src/util.cpp:136-193 -> (agCxx, agcx, agCuu, agcu, agCxu, agc0)."""
    n, m, T = dm_state, dm_act, nb_steps
    a = list(map(_f, (Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha)))
    z = lambda *s: np.zeros(s, order="F")
    out = [z(n, n, T + 1), z(n, T + 1), z(m, m, T + 1), z(m, T + 1), z(n, m, T + 1), z(T + 1)]
    rc = lib().oracle_gps_augment_cost(*map(_p, a), n, m, T, int(long_double), *map(_p, out))
    if rc:
        raise np.linalg.LinAlgError("inv_sympd(): sigma_ctl not positive definite at step %d" % (rc - 1))
    return tuple(out)


def gps_forward_pass(mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, dm_state, dm_act, nb_steps,
                     long_double=False):
    """gps/src/util.cpp:195-269 -> (mu_x, sigma_x, mu_u, sigma_u, mu_xu, sigma_xu)."""
    n, m, T = dm_state, dm_act, nb_steps
    a = list(map(_f, (mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl)))
    z = lambda *s: np.zeros(s, order="F")
    out = [z(n, T + 1), z(n, n, T + 1), z(m, T), z(m, m, T), z(n + m, T + 1), z(n + m, n + m, T + 1)]
    lib().oracle_gps_forward_pass(*map(_p, a), n, m, T, int(long_double), *map(_p, out))
    return tuple(out)


# ---- A8 / A3 / A4 / A5 / A10 ----------------------------------------------------------------------
def env_dynamics(e, x, u):
    x, u = _f(x), _f(np.atleast_1d(u))
    xn = np.zeros(e.dm_state)
    lib().oracle_env_dynamics(C.byref(e), _p(x), _p(u), _p(xn))
    return xn


def env_cost(e, x, u, u_last, a):
    x, u, ul = _f(x), _f(np.atleast_1d(u)), _f(np.atleast_1d(u_last))
    return float(lib().oracle_env_cost(C.byref(e), _p(x), _p(u), _p(ul), C.c_double(a)))


def env_jacobians(e, x, u):
    n, m = e.dm_state, e.dm_act
    x, u = _f(x), _f(np.atleast_1d(u))
    A, B = np.zeros((n, n), order="F"), np.zeros((n, m), order="F")
    if lib().oracle_env_jacobians(C.byref(e), _p(x), _p(u), _p(A), _p(B)):
        raise ValueError("unsupported (n, m)")
    return A, B


def forward_pass(e, T, x0, K, kff, alpha, xref, uref, weighting):
    n, m = e.dm_state, e.dm_act
    x0, K, kff, xref, uref, w = map(_f, (x0, K, kff, xref, uref, weighting))
    state, action, cost = np.zeros((n, T + 1), order="F"), np.zeros((m, T), order="F"), np.zeros(T + 1)
    lib().oracle_forward_pass(C.byref(e), T, _p(x0), _p(K), _p(kff), C.c_double(alpha), _p(xref), _p(uref), _p(w),
                              _p(state), _p(action), _p(cost))
    return state, action, cost


def quadratize(e, T, xref, uref, weighting, riccati=False):
    n, m = e.dm_state, e.dm_act
    xref, uref, w = map(_f, (xref, uref, weighting))
    z = lambda *s: np.zeros(s, order="F")
    A, B, c = z(n, n, T), z(n, m, T), z(n, T)
    Cxx, cx, Cuu, cu, Cxu = z(n, n, T + 1), z(n, T + 1), z(m, m, T + 1), z(m, T + 1), z(n, m, T + 1)
    if riccati:
        rc = lib().oracle_quadratize_riccati(C.byref(e), T, _p(xref), _p(uref), _p(w), _p(A), _p(B), _p(c),
                                             _p(Cxx), _p(cx), _p(Cuu), _p(cu), _p(Cxu))
    else:
        rc = lib().oracle_quadratize(C.byref(e), T, _p(xref), _p(uref), _p(w), _p(A), _p(B), _p(Cxx), _p(cx),
                                     _p(Cuu), _p(cu), _p(Cxu))
    if rc:
        raise ValueError("unsupported (n, m)")
    return dict(A=A, B=B, c=c, Cxx=Cxx, cx=cx, Cuu=Cuu, cu=cu, Cxu=Cxu)


DEFAULT_ALPHAS = np.power(10., np.linspace(0, -3, 11))  # ilqr.py:15


def ilqr_run(e, T, x0, kff_init, weighting=None, alphas=DEFAULT_ALPHAS, nb_iter=25, lmbda=1., dlmbda=1.,
             min_lmbda=1e-6, max_lmbda=1e6, mult_lmbda=1.6, tolfun=1e-6, tolgrad=1e-4, min_imp=0., reg=1):
    """ilqr.py:136-237 for one problem; returns a dict."""
    n, m = e.dm_state, e.dm_act
    w = np.ones(T + 1) if weighting is None else _f(weighting)
    x0, kff_init, alphas = _f(x0), _f(kff_init), _f(alphas)
    assert kff_init.shape == (m, T)
    opts = _f([lmbda, dlmbda, min_lmbda, max_lmbda, mult_lmbda, tolfun, tolgrad, min_imp])
    z = lambda *s: np.zeros(s, order="F")
    xref, uref, K, kff, V, v = z(n, T + 1), z(m, T), z(m, n, T), z(m, T), z(n, n, T + 1), z(n, T + 1)
    trace, scal = np.zeros(nb_iter + 1), np.zeros(7)
    lib().oracle_ilqr_run(C.byref(e), T, _p(x0), _p(kff_init), _p(w), _p(alphas), len(alphas), _p(opts), int(reg),
                          int(nb_iter), _p(xref), _p(uref), _p(K), _p(kff), _p(V), _p(v), _p(trace), _p(scal))
    return dict(xref=xref, uref=uref, K=K, kff=kff, V=V, v=v, trace=trace[:int(scal[6])].copy(),
                last_return=scal[0], lmbda=scal[1], dlmbda=scal[2], alpha=scal[3], iters=int(scal[4]),
                status=int(scal[5]))


# ---- cpu_baseline legs ----------------------------------------------------------------------------
def bench_backward_pass(stack, lmbda, reg, n, m, T, nb_problems, nb_threads):
    """`stack`: dict of problem-major stacks (nb_problems reference-layout problems back to back)."""
    K = np.zeros(nb_problems * m * n * T)
    kff = np.zeros(nb_problems * m * T)
    a = {k: np.ascontiguousarray(stack[k], dtype=np.float64).reshape(-1) for k in
         ("Cxx", "cx", "Cuu", "cu", "Cxu", "A", "B")}
    sec = lib().oracle_bench_backward_pass(_p(a["Cxx"]), _p(a["cx"]), _p(a["Cuu"]), _p(a["cu"]), _p(a["Cxu"]),
                                           _p(a["A"]), _p(a["B"]), C.c_double(lmbda), int(reg), n, m, T,
                                           C.c_int64(nb_problems), int(nb_threads), _p(K), _p(kff))
    return float(sec), K, kff


def bench_ilqr(e, T, x0s, kff_inits, nb_iter, nb_threads, weighting=None, alphas=DEFAULT_ALPHAS, opts=None,
               reg=1):
    """x0s (N, n) C-order, kff_inits (N, m*T) with each row a column-major (m, T)."""
    x0s = np.ascontiguousarray(x0s, dtype=np.float64)
    kff_inits = np.ascontiguousarray(kff_inits, dtype=np.float64)
    N = x0s.shape[0]
    w = np.ones(T + 1) if weighting is None else _f(weighting)
    alphas = _f(alphas)
    o = _f([1., 1., 1e-6, 1e6, 1.6, 1e-6, 1e-4, 0.] if opts is None else opts)
    ret = np.zeros(N)
    iters = C.c_int64(0)
    sec = lib().oracle_bench_ilqr(C.byref(e), T, _p(x0s), _p(kff_inits), _p(w), _p(alphas), len(alphas), _p(o),
                                  int(reg), int(nb_iter), C.c_int64(N), int(nb_threads), _p(ret), C.byref(iters))
    return float(sec), ret, int(iters.value)
