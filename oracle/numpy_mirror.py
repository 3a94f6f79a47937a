"""ORACLE - TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/riccati_oracle.hpp).

Independent NumPy mirror of the hot path, written from the reference's formulas rather than from
oracle/*.hpp, so that the C++ oracle is cross-checked by a second implementation:

  * backward_pass          /root/reference/trajopt/ilqr/src/util.cpp:83-182
  * riccati_backward_pass  /root/reference/trajopt/riccati/riccati.py:100-126
  * env dynamics / cost    /root/reference/trajopt/envs/{pendulum,cartpole,double_cartpole,lqr}/...
  * Jacobians by COMPLEX-STEP differentiation (exact to round-off for analytic functions), an
    independent check of the dual-number derivatives that stand in for autograd
    (/root/reference/trajopt/ilqr/objects.py:108-117)
  * forward_pass / run     /root/reference/trajopt/ilqr/ilqr.py:89-102,136-237

Pure-Python loops: small cases only.
"""
import numpy as np

EPS = np.finfo(np.float64).eps


# ---- Armadillo pieces (published algorithm, version unpinned by the reference) --------------------
def arma_is_sympd(X):
    X = np.array(X, dtype=float)
    if X.size == 0 or np.isnan(X).any():
        return False
    tol = 100.0 * EPS * np.linalg.norm(X, "fro")
    na = np.abs(X).sum(axis=1).max()
    if na != 0.0 and not (np.abs(X - X.T).sum(axis=1).max() / na <= tol):
        return False
    Y = X - tol * np.eye(X.shape[0])
    # unpivoted Cholesky: all pivots must be > 0
    n = Y.shape[0]
    L = np.zeros_like(Y)
    for j in range(n):
        d = Y[j, j] - L[j, :j] @ L[j, :j]
        if not d > 0:
            return False
        L[j, j] = np.sqrt(d)
        L[j + 1:, j] = (Y[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return True


def backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, n, m, T):
    Qxx, Qux, Quu = np.zeros((n, n, T)), np.zeros((m, n, T)), np.zeros((m, m, T))
    qx, qu = np.zeros((n, T)), np.zeros((m, T))
    V, v, dV = np.zeros((n, n, T + 1)), np.zeros((n, T + 1)), np.zeros(2)
    K, kff = np.zeros((m, n, T)), np.zeros((m, T))
    diverge, failed = 0, False
    V[..., T], v[..., T] = Cxx[..., T], cx[..., T]
    for i in range(T - 1, -1, -1):
        a, b, Vn, vn = A[..., i], B[..., i], V[..., i + 1], v[..., i + 1]
        Qxx[..., i] = Cxx[..., i] + a.T @ Vn @ a
        Quu[..., i] = Cuu[..., i] + b.T @ Vn @ b
        Qux[..., i] = (Cxu[..., i] + a.T @ Vn @ b).T
        qu[..., i] = cu[..., i] + b.T @ vn
        qx[..., i] = cx[..., i] + a.T @ vn
        Vreg = Vn + (lmbda * np.eye(n) if reg == 2 else 0.0)
        Qux_reg = (Cxu[..., i] + a.T @ Vreg @ b).T
        Quu_reg = Cuu[..., i] + b.T @ Vreg @ b + (lmbda * np.eye(m) if reg == 1 else 0.0)
        if not arma_is_sympd(Quu_reg):
            diverge, failed = i, True
            break
        Quu_inv = np.linalg.inv(Quu_reg)
        K[..., i] = -Quu_inv @ Qux_reg
        kff[..., i] = -Quu_inv @ qu[..., i]
        dV += np.array([kff[..., i] @ qu[..., i], 0.5 * kff[..., i] @ Quu[..., i] @ kff[..., i]])
        v[..., i] = qx[..., i] + K[..., i].T @ Quu[..., i] @ kff[..., i] + K[..., i].T @ qu[..., i] \
            + Qux[..., i].T @ kff[..., i]
        Vi = Qxx[..., i] + K[..., i].T @ Quu[..., i] @ K[..., i] + K[..., i].T @ Qux[..., i] \
            + Qux[..., i].T @ K[..., i]
        V[..., i] = 0.5 * (Vi + Vi.T)
    return (Qxx, Qux, Quu, qx, qu, V, v, dV, K, kff, diverge), failed


def riccati_backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, c, n, m, T, literal_elementwise=False):
    V, v = np.zeros((n, n, T + 1)), np.zeros((n, T + 1))
    K, kff = np.zeros((m, n, T)), np.zeros((m, T))
    V[..., T], v[..., T] = Cxx[..., T], cx[..., T]
    for t in range(T - 1, -1, -1):
        a, b, ct, Vn, vn = A[..., t], B[..., t], c[..., t], V[..., t + 1], v[..., t + 1]
        Qxx = Cxx[..., t] + a.T @ Vn @ a
        Quu = Cuu[..., t] + b.T @ Vn @ b
        Qux = Cxu[..., t].T + b.T @ Vn @ a
        qx = cx[..., t] + 2.0 * a.T @ Vn @ ct + a.T @ vn
        qu = cu[..., t] + 2.0 * b.T @ Vn @ ct + b.T @ vn
        Quu_inv = np.linalg.inv(Quu)
        K[..., t] = -Quu_inv @ Qux
        kff[..., t] = -0.5 * Quu_inv @ qu
        # riccati.py:123 writes an elementwise product; it equals Qux^T K only for m == 1
        V[..., t] = Qxx + (Qux.T * K[..., t] if literal_elementwise else Qux.T @ K[..., t])
        v[..., t] = qx + 2.0 * kff[..., t].T @ Qux
    return K, kff, V, v


def gps_backward_pass(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha, n, m, T):
    """gps/src/util.cpp:272-374 with NumPy's own inv / det.  -> dict of the 12 arrays + diverge."""
    o = dict(Qxx=np.zeros((n, n, T)), Qux=np.zeros((m, n, T)), Quu=np.zeros((m, m, T)), qx=np.zeros((n, T)),
             qu=np.zeros((m, T)), q0=np.zeros(T), V=np.zeros((n, n, T + 1)), v=np.zeros((n, T + 1)),
             v0=np.zeros(T + 1), K=np.zeros((m, n, T)), kff=np.zeros((m, T)), sigma_ctl=np.zeros((m, m, T)), diverge=0)
    o["V"][..., T], o["v"][..., T], o["v0"][T] = Cxx[..., T], cx[..., T], c0[T]
    for i in range(T - 1, -1, -1):
        a, b, ci, Vn, vn, al = A[..., i], B[..., i], c[..., i], o["V"][..., i + 1], o["v"][..., i + 1], alpha[i]
        Qxx = -(Cxx[..., i] + a.T @ Vn @ a) / al
        Quu = -(Cuu[..., i] + b.T @ Vn @ b) / al
        Qux = -(Cxu[..., i] + a.T @ Vn @ b).T / al
        qu = -(cu[..., i] + 2.0 * b.T @ Vn @ ci + b.T @ vn) / al
        qx = -(cx[..., i] + 2.0 * a.T @ Vn @ ci + a.T @ vn) / al
        q0 = -(c0[i] + o["v0"][i + 1] + ci @ Vn @ ci + np.trace(Vn @ sigma_dyn[..., i]) + vn @ ci) / al
        o["Qxx"][..., i], o["Quu"][..., i], o["Qux"][..., i] = Qxx, Quu, Qux
        o["qu"][..., i], o["qx"][..., i], o["q0"][i] = qu, qx, q0
        if np.allclose(Quu, Quu.T, rtol=0, atol=1e-12 * np.abs(Quu).max()) and np.all(np.linalg.eigvalsh(Quu) > 0):
            o["diverge"] = i
            break
        Qi = np.linalg.inv(Quu)
        K, kff = -Qi @ Qux, -0.5 * Qi @ qu
        sg = -0.5 * Qi
        o["K"][..., i], o["kff"][..., i], o["sigma_ctl"][..., i] = K, kff, 0.5 * (sg.T + sg)
        Vi = -al * (Qxx + Qux.T @ K)
        o["V"][..., i] = 0.5 * (Vi + Vi.T)
        o["v"][..., i] = -al * (qx + 2.0 * Qux.T @ kff)
        o["v0"][i] = -al * (0.5 * qu @ kff + q0 + 0.5 * (m * np.log(2.0 * np.pi) - np.log(np.linalg.det(-2.0 * Quu))))
    return o


# ---- env models (work on float or complex inputs) ----------------------------------------------------
def _clip(x, lim):
    """np.clip(x, -lim, lim) on the real part; the imaginary part follows autograd's mask."""
    x = np.asarray(x)
    lim = np.broadcast_to(np.asarray(lim, dtype=float), x.shape)
    r = np.minimum(np.maximum(x.real, -lim), lim)
    if np.iscomplexobj(x):
        mask = (r != -lim) & (r != lim)
        return r + 1j * np.where(mask, x.imag, 0.0)
    return r


def _quad_pendulum_f(x, u):
    """Four-link planar chain from first principles (an independent derivation, not the reference's expanded
    expressions): link k has absolute angle phi_k = th_1 + pi + th_2 + ... + th_k, mass 1, length 1, centre of mass at
    1/2, inertia m (l^2 + 1e-4) / 3 about the centre as quad_pendulum.py:74 defines it; joint friction 0.025.
    T = 1/2 sum m |v_ck|^2 + 1/2 sum I_k dphi_k^2, V = g sum m y_ck (th_1 = 0 upright); Euler-Lagrange:
    M ddq + (sum_i dq_i dM/dq_i) dq - 1/2 [dq' dM/dq_k dq]_k + dV/dq = u - friction dq.
    Written with analytic derivatives only, so that it stays holomorphic for the complex-step Jacobians."""
    g = 9.81
    m, l, fr = np.ones(4), np.ones(4), 0.025 * np.ones(4)
    I = m * (l ** 2 + 1e-4) / 3.0
    x = np.asarray(x)
    q, dq = x[:4], x[4:]
    cum = np.tril(np.ones((4, 4)))                      # phi = cum @ th (+ pi)
    phi = cum @ q + np.pi
    sp, cp = np.sin(phi), np.cos(phi)
    W = np.array([[l[j] if j < k else (0.5 * l[k] if j == k else 0.0) for j in range(4)] for k in range(4)])
    # centre k: sum_j W[k, j] (sin phi_j, -cos phi_j); velocity Jacobians w.r.t. th
    Jx = (W * cp[None, :]) @ cum
    Jy = (W * sp[None, :]) @ cum
    Mq = sum(m[k] * (np.outer(Jx[k], Jx[k]) + np.outer(Jy[k], Jy[k])) + I[k] * np.outer(cum[k], cum[k]) for k in range(4))
    dM = []
    for i in range(4):
        dJx = (W * (-sp * cum[:, i])[None, :]) @ cum       # d phi_j / d th_i = cum[j, i]
        dJy = (W * (cp * cum[:, i])[None, :]) @ cum
        dM.append(sum(m[k] * (np.outer(dJx[k], Jx[k]) + np.outer(Jx[k], dJx[k]) + np.outer(dJy[k], Jy[k])
                              + np.outer(Jy[k], dJy[k])) for k in range(4)))
    dV = np.array([g * np.sum(m * (W @ (sp * cum[:, i]))) for i in range(4)])   # y_ck = -sum_j W[k, j] cos phi_j
    cor = sum(dq[i] * dM[i] for i in range(4)) @ dq - 0.5 * np.array([dq @ dM[k] @ dq for k in range(4)])
    dd = np.linalg.solve(Mq, np.asarray(u) - fr * dq - cor - dV)
    return np.concatenate([dq, dd])


class Env:
    """Parameter bag with the reference's constants; `name` selects the vector field."""

    def __init__(self, name, dt=None, uw=None, slew_rate=False, periodic=False, cartesian=False, **kw):
        self.features = None
        inf = np.inf
        lqr_dt = 0.1
        if name in ("lqr_v1", "lqr_v2"):   # lqr_v1.py:10-37, lqr_v2.py:10-36: other constants, same affine model
            lqr_dt = 0.01 if dt is None else dt
            if name == "lqr_v1":
                mm, k, d = 1., 1e-2, 1e-1
                kw = dict(A=np.array([[0., 1.], [-k / mm, -d / mm]]) * 0.01 + np.eye(2),   # the constructor's dt
                          B=np.array([[0.], [1. / mm]]) * 0.01, gw=[1e2, 1e0])
            else:
                kw = dict(A=[[1.0, 0.], [0.1, 1.1]], B=[[0.05], [0.]], gw=[1e1, 1e0])
            kw.update(goal=[1., 0.], c=[0., 0.])
            uw = [1e-3] if uw is None else uw
            name = "lqr"
        self.name, self.slew_rate, self.periodic = name, slew_rate, periodic
        self.cost_times_dt = False
        if name == "pendulum":
            self.n, self.m, self.dt = 2, 1, 0.01
            self.umax, self.xmax = np.array([10.]), np.array([inf, inf])
            self.g, self.gw, self.uw = np.zeros(2), np.array([1e0, 1e-1]), np.array([1e-3])
            self.angles = [0]
        elif name == "cartpole":
            self.n, self.m, self.dt = 4, 1, 0.01
            self.umax, self.xmax = np.array([10.]), np.array([10., inf, inf, inf])
            self.g, self.gw, self.uw = np.zeros(4), np.array([1e1, 1e4, 1e0, 1e0]), np.array([1e-5])
            self.angles = [1]
        elif name == "double_cartpole":
            self.n, self.m, self.dt = 6, 1, 0.01
            self.umax, self.xmax = np.array([5.]), np.array([10.] + [inf] * 5)
            self.g, self.gw, self.uw = np.zeros(6), np.array([1e1, 1e4, 1e4, 1e0, 1e0, 1e0]), np.array([1e-5])
            self.angles = [1, 2]
        elif name == "double_pendulum":
            self.n, self.m, self.dt = 4, 2, 0.01
            self.umax, self.xmax = np.array([10., 10.]), np.full(4, inf)
            self.g, self.gw, self.uw = np.zeros(4), np.array([1e4, 1e4, 1e0, 1e0]), np.full(2, 1e-5)
            self.angles = [0]
        elif name == "lqr":
            self.A = np.asarray(kw.get("A", [[1.1, 0.], [1.0, 1.0]]), dtype=float)
            self.B = np.asarray(kw.get("B", [[1.], [0.]]), dtype=float)
            self.n, self.m = self.B.shape
            self.dt = lqr_dt
            self.g = np.asarray(kw.get("goal", [10., 10.]), dtype=float)
            self.c = np.asarray(kw.get("c", -self.A @ self.g), dtype=float)
            self.gw, self.uw = np.asarray(kw.get("gw", [1e1] * self.n), dtype=float), np.ones(self.m)
            self.umax, self.xmax = np.full(self.m, inf), np.full(self.n, inf)
            self.cost_times_dt = True
            self.angles = []
        elif name == "quad_pendulum":
            self.n, self.m, self.dt = 8, 4, 0.01
            self.umax, self.xmax = np.full(4, 25.), np.full(8, inf)
            self.g, self.gw, self.uw = np.zeros(8), np.array([1e4] * 4 + [1e0] * 4), np.full(4, 1e-5)
            self.angles = [0, 1, 2, 3]
        elif name == "robot":
            self.n, self.m, self.dt = 4, 2, 0.025
            self.umax, self.xmax = np.full(2, inf), np.full(4, inf)
            self.g, self.gw, self.uw = np.zeros(4), np.array([1e1, 1e1, 1., 1.]), np.full(2, 1e-3)
            self.cost_times_dt = True
            self.angles = []
        elif name == "quadrotor":
            self.n, self.m, self.dt = 12, 4, 0.01
            self.umax, self.xmax = np.full(4, 5.), np.full(12, inf)
            self.g = np.array([1., 1., 1.] + [0.] * 9)
            self.gw = np.array([1e1] * 3 + [1e0] * 3 + [1e0] * 3 + [1e-1] * 3)
            self.uw = np.full(4, 1e-3)
            self.angles = []
        else:
            raise KeyError(name)
        if cartesian:   # the "...WithCartesianCost" classes, written out as the reference writes them
            self.features = {
                "pendulum": lambda x: np.array([np.cos(x[0]), np.sin(x[0]), x[1]]),
                "cartpole": lambda x: np.array([x[0], np.cos(x[1]), np.sin(x[1]), x[2], x[3]]),
                "double_cartpole": lambda x: np.array([x[0], np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]),
                                                       x[2], x[3], x[4]]),   # double_cartpole.py:226-230, as written
                "double_pendulum": lambda x: np.array([np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]),
                                                       x[2], x[3]]),
                "quad_pendulum": lambda x: np.array([np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]),
                                                     np.cos(x[2]), np.sin(x[2]), np.cos(x[3]), np.sin(x[3]),
                                                     x[4], x[5], x[6], x[7]]),
            }[name]
            self.g, self.gw = {
                "pendulum": (np.array([1., 0., 0.]), np.array([1e4, 1e4, 1e0])),
                "cartpole": (np.array([0., 1., 0., 0., 0.]), np.array([1e1, 1e4, 1e4, 1e0, 1e0])),
                "double_cartpole": (np.array([0., 1., 0., 1., 0., 0., 0., 0.]),
                                    np.array([1e1, 1e4, 1e4, 1e4, 1e4, 1e0, 1e0, 1e0])),
                "double_pendulum": (np.array([1., 0., 1., 0., 0., 0.]), np.array([1e4, 1e4, 1e4, 1e4, 1e0, 1e0])),
                "quad_pendulum": (np.array([1., 0.] * 4 + [0.] * 4), np.array([1e4] * 8 + [1e0] * 4)),
            }[name]
        if dt is not None:
            self.dt = dt
        if uw is not None:
            self.uw = np.asarray(uw, dtype=float) * np.ones(self.m)

    # continuous-time vector fields
    def f(self, x, u):
        nm = self.name
        if nm == "pendulum":
            g, m, l, k = 9.81, 1., 1., 1e-3
            th, dth = x
            return np.array([dth, -3. * g / (2. * l) * np.sin(th + np.pi) + 3. / (m * l ** 2) * (u[0] - k * dth)])
        if nm == "cartpole":
            g, Mc, Mp, l, fr = 9.81, 0.37, 0.127, 0.3365, 0.005
            Mt = Mc + Mp
            q, th, dq, dth = x
            sth, cth = np.sin(th), np.cos(th)
            num = g * sth + cth * (-(u[0] - fr * dq) - Mp * l * dth ** 2 * sth) / Mt
            denom = l * ((4. / 3.) - Mp * cth ** 2 / Mt)
            ddth = num / denom
            ddx = (u[0] + Mp * l * (dth ** 2 * sth - ddth * cth)) / Mt
            return np.array([dq, dth, ddx, ddth])
        if nm == "double_cartpole":
            g, Mc, Mp1, Mp2, L1, L2 = 9.81, 0.37, 0.127, 0.127, 0.3365, 0.3365
            Mt, l1, l2 = Mc + Mp1 + Mp2, L1 / 2., L2 / 2.
            J1, J2 = Mp1 * L1 / 12., Mp2 * L2 / 12.
            q, th1, th2, dq, dth1, dth2 = x
            s1, c1, s2, c2 = np.sin(th1), np.cos(th1), np.sin(th2), np.cos(th2)
            sd, cd = np.sin(th1 - th2), np.cos(th1 - th2)
            a = Mp1 * l1 + Mp2 * L2
            b = Mp2 * l2
            d = L1 * l2 * Mp2
            M = np.array([[Mt, a * c1, b * c2],
                          [a * c1, l1 ** 2 * Mp1 + L1 ** 2 * Mp2 + J1, d * cd],
                          [b * c2, d * cd, l2 ** 2 * Mp2 + J2]])
            Cm = np.array([[0., -a * dth1 * s1, -b * dth2 * s2],
                           [0., 0., d * dth2 * sd],
                           [0., -d * dth1 * sd, 0.]])
            G = np.array([0., -(Mp1 * l1 + Mp2 * L1) * g * s1, -Mp2 * l2 * g * s2])
            act = np.array([u[0], 0., 0.])
            dd = np.linalg.solve(M, act - Cm @ np.array([dq, dth1, dth2]) - G)
            return np.concatenate([np.array([dq, dth1, dth2]), dd])
        if nm == "double_pendulum":
            g, m1, m2, l1, l2, k1, k2 = 9.81, 1., 1., 1., 1., 0.025, 0.025
            I1, I2 = m1 * (l1 ** 2 + 1e-4) / 3., m2 * (l2 ** 2 + 1e-4) / 3.
            a1, a2 = l1 / 2., l2 / 2.
            th1, th2, dth1, dth2 = x[0] + np.pi, x[1], x[2], x[3]
            s1, c1, s2, c2 = np.sin(th1), np.cos(th1), np.sin(th2), np.cos(th2)
            H = np.array([[I1 + I2 + a1 ** 2 * m1 + l1 ** 2 * m2 + a2 ** 2 * m2 + 2. * l1 * a2 * m2 * c2,
                           I2 + a2 ** 2 * m2 + l1 * a2 * m2 * c2],
                          [I2 + a2 ** 2 * m2 + l1 * a2 * m2 * c2, I2 + a2 ** 2 * m2]])
            b = np.array([g * a1 * m1 * s1 + g * l1 * m2 * s1 + g * a2 * m2 * c2 * s1
                          - 2. * dth1 * dth2 * l1 * a2 * m2 * s2 - dth2 ** 2 * l1 * a2 * m2 * s2 + g * a2 * m2 * c1 * s2,
                          g * a2 * m2 * c2 * s1 + dth1 ** 2 * l1 * a2 * m2 * s2 + g * a2 * m2 * c1 * s2])
            tau = np.array([u[0] - k1 * dth1, u[1] - k2 * dth2])
            dd = np.linalg.solve(H, tau - b)    # (h22 r1 - h12 r2) / det etc. is Cramer's rule for this system
            return np.array([dth1, dth2, dd[0], dd[1]])
        if nm == "quadrotor":
            g, mass, L, Ixx, Iyy, Izz, kM = 9.81, 0.5, 0.25, 3.9e-3, 3.9e-3, 7.5e-3, 1.6e-2
            ph, th, ps = x[3], x[4], x[5]
            p, q, r = x[9], x[10], x[11]
            F = u[0] + u[1] + u[2] + u[3]
            tx, ty, tz = L * (u[1] - u[3]), L * (u[2] - u[0]), kM * (u[0] - u[1] + u[2] - u[3])
            sph, cph, sth, cth, sps, cps = np.sin(ph), np.cos(ph), np.sin(th), np.cos(th), np.sin(ps), np.cos(ps)
            tth = sth / cth
            return np.array([x[6], x[7], x[8],
                             p + sph * tth * q + cph * tth * r, cph * q - sph * r, (sph * q + cph * r) / cth,
                             (cph * sth * cps + sph * sps) * F / mass, (cph * sth * sps - sph * cps) * F / mass,
                             cph * cth * F / mass - g,
                             (tx - (Izz - Iyy) * q * r) / Ixx, (ty - (Ixx - Izz) * p * r) / Iyy,
                             (tz - (Iyy - Ixx) * p * q) / Izz])
        if nm == "quad_pendulum":
            return _quad_pendulum_f(x, u)
        if nm == "robot":
            return np.array([x[3] * np.cos(x[2]), x[3] * np.sin(x[2]), x[3] * np.tan(u[1]) / 0.1, u[0]])
        raise KeyError(nm)

    def dynamics(self, x, u):
        x, u = np.asarray(x), np.atleast_1d(np.asarray(u))
        _u = _clip(u, self.umax)
        if self.name == "lqr":
            return self.A @ x + self.B @ _u + self.c
        dt = -self.dt if self.name == "robot" else self.dt   # robot.py:62-67
        k1 = self.f(x, _u)
        k2 = self.f(x + 0.5 * dt * k1, _u)
        k3 = self.f(x + 0.5 * dt * k2, _u)
        k4 = self.f(x + dt * k3, _u)
        return _clip(x + dt / 6. * (k1 + 2. * k2 + 2. * k3 + k4), self.xmax)

    def cost(self, x, u, u_last, a, frozen_at=None):
        """frozen_at: the point whose feature Jacobian is used (getval(y), cartpole.py:122-124); None = x itself."""
        x, u, u_last = np.asarray(x, dtype=float), np.atleast_1d(u).astype(float), np.atleast_1d(u_last).astype(float)
        d = (u - u_last) if self.slew_rate else u
        c = d @ np.diag(self.uw) @ d
        if a:
            y = x.copy()
            if self.periodic:
                for i in self.angles:
                    y[i] = ((x[i] + np.pi) % (2. * np.pi)) - np.pi
            if self.features is not None:
                y0 = y if frozen_at is None else np.asarray(frozen_at, dtype=float)
                hh = 1e-30   # complex-step Jacobian of the features at the frozen point
                J = np.stack([self.features(y0 + 1j * hh * np.eye(self.n)[k]).imag / hh for k in range(self.n)], axis=1)
                z = J @ y + (self.features(y0) - J @ y0)
            else:
                z = y
            c += a * (z - self.g) @ np.diag(self.gw) @ (z - self.g)
        return self.dt * c if self.cost_times_dt else c

    # derivatives
    def jacobians(self, x, u, h=1e-30):
        """Complex-step d dynamics/dx, d dynamics/du."""
        x, u = np.asarray(x, dtype=float), np.atleast_1d(np.asarray(u, dtype=float))
        A, B = np.zeros((self.n, self.n)), np.zeros((self.n, self.m))
        for j in range(self.n):
            xp = x.astype(complex)
            xp[j] += 1j * h
            A[:, j] = self.dynamics(xp, u.astype(complex)).imag / h
        for j in range(self.m):
            up = u.astype(complex)
            up[j] += 1j * h
            B[:, j] = self.dynamics(x.astype(complex), up).imag / h
        return A, B

    def cost_derivatives_fd(self, x, u, u_last, a, h=1.0):
        """Central differences of the literal cost; exact for any h because the cost is quadratic, and h = 1
        keeps the round-off of the second differences at the 1e-16 relative level."""
        x, u, u_last = np.asarray(x, dtype=float), np.atleast_1d(u).astype(float), np.atleast_1d(u_last).astype(float)
        z0 = np.concatenate([x, u])
        n, m = self.n, self.m

        def c(z):
            return self.cost(z[:n], z[n:], u_last, a, frozen_at=x if self.features is not None else None)

        N = n + m
        g, H = np.zeros(N), np.zeros((N, N))
        I = np.eye(N) * h
        for i in range(N):
            g[i] = (c(z0 + I[i]) - c(z0 - I[i])) / (2 * h)
            for j in range(N):
                H[i, j] = (c(z0 + I[i] + I[j]) - c(z0 + I[i] - I[j]) - c(z0 - I[i] + I[j]) + c(z0 - I[i] - I[j])) / (4 * h * h)
        return H[:n, :n], g[:n], H[n:, n:], g[n:], H[:n, n:]


def forward_pass(env, T, x0, K, kff, alpha, xref, uref, weighting):
    n, m = env.n, env.m
    state, action, cost = np.zeros((n, T + 1)), np.zeros((m, T)), np.zeros(T + 1)
    state[:, 0] = x0
    for t in range(T):
        a = uref[:, t] + alpha * kff[:, t] + K[..., t] @ (state[:, t] - xref[:, t])
        action[:, t] = np.clip(a, -env.umax, env.umax)
        cost[t] = env.cost(state[:, t], action[:, t], action[:, t], weighting[t])
        state[:, t + 1] = env.dynamics(state[:, t], action[:, t])
    cost[T] = env.cost(state[:, T], np.zeros(m), np.zeros(m), weighting[T])
    return state, action, cost


def quadratize(env, T, xref, uref, weighting):
    """Closed forms of SURVEY 9.1 for the cost (identity features), complex-step for the dynamics."""
    n, m = env.n, env.m
    A, B = np.zeros((n, n, T)), np.zeros((n, m, T))
    Cxx, cx = np.zeros((n, n, T + 1)), np.zeros((n, T + 1))
    Cuu, cu, Cxu = np.zeros((m, m, T + 1)), np.zeros((m, T + 1)), np.zeros((n, m, T + 1))
    upad = np.hstack([uref, np.zeros((m, 1))])
    for t in range(T):
        A[..., t], B[..., t] = env.jacobians(xref[:, t], uref[:, t])
    for t in range(T + 1):
        Cxx[..., t], cx[:, t], Cuu[..., t], cu[:, t], Cxu[..., t] = env.cost_derivatives_fd(
            xref[:, t], upad[:, t], upad[:, t - 1], weighting[t])
    return dict(A=A, B=B, Cxx=Cxx, cx=cx, Cuu=Cuu, cu=cu, Cxu=Cxu)


def ilqr_run(env, T, x0, kff_init, weighting, alphas, nb_iter, lmbda=1., dlmbda=1., min_lmbda=1e-6, max_lmbda=1e6,
             mult_lmbda=1.6, tolfun=1e-6, tolgrad=1e-4, min_imp=0., reg=1, quad=None):
    """Driver mirror (ilqr.py:136-237).  `quad` lets a test substitute the quadratization."""
    n, m = env.n, env.m
    quad = quad or quadratize
    xref, uref = np.zeros((n, T + 1)), np.zeros((m, T))
    xref[:, 0] = x0
    K, kff = np.zeros((m, n, T)), np.array(kff_init, dtype=float)
    last_return, trace = -np.inf, []
    for alpha in alphas:
        st, ac, co = forward_pass(env, T, x0, K, kff, alpha, xref, uref, weighting)
        if np.all(st < 1e8):
            xref, uref, last_return = st, ac, co.sum()
            break
    trace.append(last_return)
    status = 0
    for _ in range(nb_iter):
        q = quad(env, T, xref, uref, weighting)
        done = False
        while not done:
            out, failed = backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], lmbda, reg,
                                        n, m, T)
            if failed:
                dlmbda = max(dlmbda * mult_lmbda, mult_lmbda)
                lmbda = max(lmbda * dlmbda, min_lmbda)
                if lmbda > max_lmbda:
                    break
            else:
                done = True
        lK, lkff, dV = out[8], out[9], out[7]
        g_norm = np.mean(np.max(np.abs(lkff) / (np.abs(uref) + 1.), axis=1))
        if g_norm < tolgrad and lmbda < 1e-5:
            dlmbda = min(dlmbda / mult_lmbda, 1. / mult_lmbda)
            lmbda = lmbda * dlmbda * (lmbda > min_lmbda)
            status = 1
            break
        accepted = False
        if done:
            for alpha in alphas:
                st, ac, co = forward_pass(env, T, x0, lK, lkff, alpha, xref, uref, weighting)
                ret = co.sum()
                dret = last_return - ret
                with np.errstate(all="ignore"):
                    imp = np.float64(dret) / np.float64(-alpha * (dV[0] + alpha * dV[1]))
                if imp >= min_imp:
                    accepted = True
                    break
        if accepted:
            dlmbda = min(dlmbda / mult_lmbda, 1. / mult_lmbda)
            lmbda = lmbda * dlmbda * (lmbda > min_lmbda)
            xref, uref, last_return, K, kff = st, ac, ret, lK, lkff
            trace.append(last_return)
            if dret < tolfun:
                status = 2
                break
        else:
            dlmbda = max(dlmbda * mult_lmbda, mult_lmbda)
            lmbda = max(lmbda * dlmbda, min_lmbda)
            if lmbda > max_lmbda:
                status = 3
                break
    return dict(xref=xref, uref=uref, K=K, kff=kff, trace=np.array(trace), last_return=last_return, lmbda=lmbda,
                dlmbda=dlmbda, status=status)


def gps_kl_divergence(p_K, p_kff, p_sigma, q_K, q_kff, q_sigma, mu_x, sigma_x, n, m, T):
    """gps/src/util.cpp:83-121 with np.linalg.inv / det."""
    kl = np.zeros(T)
    for i in range(T):
        lam = np.linalg.inv(q_sigma[..., i])
        dK, dk = q_K[..., i] - p_K[..., i], p_kff[..., i] - q_kff[..., i]
        mu = mu_x[..., i]
        kl[i] = (0.5 * np.log(np.linalg.det(q_sigma[..., i]) / np.linalg.det(p_sigma[..., i]))
                 + 0.5 * np.trace(lam @ p_sigma[..., i]) - 0.5 * m
                 + 0.5 * np.trace(dK.T @ lam @ dK @ sigma_x[..., i]) + 0.5 * mu @ dK.T @ lam @ dK @ mu
                 - mu @ dK.T @ lam @ dk + 0.5 * dk @ lam @ dk)
    return kl


def gps_quad_expectation(mu, sigma_s, Q, q, q0):
    """gps/src/util.cpp:123-134."""
    return float(mu @ Q @ mu + mu @ q + q0 + np.trace(Q @ sigma_s))


def gps_augment_cost(Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, n, m, T):
    """gps/src/util.cpp:136-193."""
    ag = [np.array(a, dtype=float, order="F") for a in (Cxx, cx, Cuu, cu, Cxu, c0)]
    for i in range(T):
        lam = np.linalg.inv(sigma_ctl[..., i])
        Ki, ki, al = K[..., i], kff[..., i], alpha[i]
        ag[0][..., i] = Cxx[..., i] + 0.5 * al * Ki.T @ lam @ Ki
        ag[2][..., i] = Cuu[..., i] + 0.5 * al * lam
        ag[4][..., i] = Cxu[..., i] - 0.5 * al * Ki.T @ lam
        ag[1][..., i] = cx[..., i] + al * Ki.T @ lam @ ki
        ag[3][..., i] = cu[..., i] - al * lam @ ki
        ag[5][i] = c0[i] + 0.5 * al * np.log(np.linalg.det(2.0 * np.pi * sigma_ctl[..., i])) + 0.5 * al * ki @ lam @ ki
    return tuple(ag)


def gps_forward_pass(mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, n, m, T):
    """gps/src/util.cpp:195-269 (the parts of the last mu_xu / sigma_xu slices the reference leaves unset are zero)."""
    mu_x, sigma_x = np.zeros((n, T + 1)), np.zeros((n, n, T + 1))
    mu_u, sigma_u = np.zeros((m, T)), np.zeros((m, m, T))
    mu_xu, sigma_xu = np.zeros((n + m, T + 1)), np.zeros((n + m, n + m, T + 1))
    mu_x[:, 0], sigma_x[..., 0] = mu_x0, sigma_x0
    sym = lambda X: 0.5 * (X + X.T)
    for i in range(T):
        Ki, sx = K[..., i], sigma_x[..., i]
        mu_u[:, i] = Ki @ mu_x[:, i] + kff[:, i]
        sigma_u[..., i] = sym(sigma_ctl[..., i] + Ki @ sx @ Ki.T)
        sigma_xu[..., i] = sym(np.block([[sx, sx @ Ki.T], [Ki @ sx, sigma_u[..., i]]]))
        mu_xu[:, i] = np.concatenate([mu_x[:, i], mu_u[:, i]])
        F = np.hstack([A[..., i], B[..., i]])
        sigma_x[..., i + 1] = sym(sigma_dyn[..., i] + F @ sigma_xu[..., i] @ F.T)
        mu_x[:, i + 1] = F @ mu_xu[:, i] + c[:, i]
    mu_xu[:n, T] = mu_x[:, T]
    sigma_xu[:n, :n, T] = sigma_x[..., T]
    return mu_x, sigma_x, mu_u, sigma_u, mu_xu, sigma_xu


def bsp_backward_pass(Q, q, R, r, P, p, F, G, T_, U, V, X, Y, Z, lmbda, reg, b, m, T, tau_mode=0):
    """bspilqr/src/util.cpp:83-204 with np.linalg.inv; tau_mode as in oracle/bsp_oracle.hpp.  -> dict."""
    S, s, tau = np.zeros((b, b, T + 1)), np.zeros((b, T + 1)), np.zeros((b * b, T + 1))
    K, kff, dS, diverge = np.zeros((m, b, T)), np.zeros((m, T)), np.zeros(2), -1
    S[..., T], s[:, T], tau[:, T] = Q[..., T], q[:, T], p[:, T]
    for i in range(T - 1, -1, -1):
        Sn, sn, tn = S[..., i + 1], s[:, i + 1], tau[:, i + 1]
        vS = Sn.flatten(order="F")
        Cm = Q[..., i] + F[..., i].T @ Sn @ F[..., i]
        D = R[..., i] + G[..., i].T @ Sn @ G[..., i]
        E = (P[..., i] + F[..., i].T @ Sn @ G[..., i]).T
        c = q[:, i] + F[..., i].T @ sn + T_[..., i].T @ tn + 0.5 * X[..., i].T @ vS
        d = r[:, i] + G[..., i].T @ sn + V[..., i].T @ tn + 0.5 * Z[..., i].T @ vS
        e = p[:, i] + (U[..., i].T @ tn if tau_mode == 0 else 0.0) + 0.5 * Y[..., i].T @ vS
        Sreg = Sn + (lmbda * np.eye(b) if reg == 2 else 0.0)
        Ereg = (P[..., i] + F[..., i].T @ Sreg @ G[..., i]).T
        Dreg = R[..., i] + G[..., i].T @ Sreg @ G[..., i] + (lmbda * np.eye(m) if reg == 1 else 0.0)
        if not arma_is_sympd(Dreg):
            diverge = i
            break
        Dinv = np.linalg.inv(Dreg)
        K[..., i], kff[:, i] = -Dinv @ Ereg, -Dinv @ d
        dS += np.array([kff[:, i] @ d, 0.5 * kff[:, i] @ D @ kff[:, i]])
        tau[:, i] = e
        s[:, i] = c + K[..., i].T @ D @ kff[:, i] + K[..., i].T @ d + E.T @ kff[:, i]
        Si = Cm + K[..., i].T @ D @ K[..., i] + K[..., i].T @ E + E.T @ K[..., i]
        S[..., i] = 0.5 * (Si + Si.T)
    return dict(S=S, s=s, tau=tau, dS=dS, K=K, kff=kff, diverge=diverge)
