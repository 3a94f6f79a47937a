"""iLQR for an env this library has no device model of ("unknown Python envs fall back to host
quadratisation feeding the GPU backward pass").

The env is any object with the reference's interface - `dynamics(x, u)`, `cost(x, u, u_last, a)`, `dm_state`, `dm_act`
(/root/reference/trajopt/ilqr/ilqr.py:23-31).  Per iteration:
  * the expansion (ilqr.py:152-156) is taken on the host by central differences of those two callables - the reference
    differentiates them with autograd, which this package does not depend on, so the derivatives carry finite-difference
    error (~1e-7 relative) where the device models' dual numbers are exact;
  * the regularised backward pass (ilqr.py:104-117) runs on the GPU through the C ABI (`core.backward_pass`);
  * rollouts and the line search (ilqr.py:89-102,187-202) run on the host, one alpha after the other, because the
    dynamics is a Python callable.
The control flow is iLQR.run's (ilqr.py:136-237): lambda schedule, first-accept step size, the three stopping rules.
It is a compatibility path, orders of magnitude slower than the device driver, and it fails loudly without a GPU."""
import numpy as np

from . import core


def _central(f, z, h):
    """Jacobian of f: R^k -> R^r at z by central differences with per-coordinate steps h * max(1, |z_i|)."""
    z = np.asarray(z, dtype=np.float64)
    cols = []
    for i in range(z.size):
        d = h * max(1.0, abs(z[i]))
        zp, zm = z.copy(), z.copy()
        zp[i] += d
        zm[i] -= d
        cols.append((np.atleast_1d(f(zp)) - np.atleast_1d(f(zm))) / (2.0 * d))
    return np.stack(cols, axis=-1)


class HostEnvSolver:
    def __init__(self, owner):
        self.o = owner            # the iLQR instance: parameters, xref / uref / ctl, weighting
        u = owner.env.unwrapped if hasattr(owner.env, "unwrapped") else owner.env
        self.dyn, self.cost = u.dynamics, u.cost
        self.n, self.m, self.T = owner.dm_state, owner.dm_act, owner.nb_steps

    # ilqr.py:89-102
    def rollout(self, ctl, alpha):
        o, n, m, T = self.o, self.n, self.m, self.T
        x, u, c = np.zeros((n, T + 1)), np.zeros((m, T)), np.zeros(T + 1)
        x[:, 0] = np.asarray(o.env_init, dtype=np.float64)
        for t in range(T):
            u[:, t] = ctl.action(x, alpha, o.xref, o.uref, t)
            u[:, t] = np.clip(u[:, t], -o.ulim, o.ulim)
            c[t] = self.cost(x[:, t], u[:, t], u[:, t], o.weighting[t])
            x[:, t + 1] = self.dyn(x[:, t], u[:, t])
        c[T] = self.cost(x[:, T], np.zeros(m), np.zeros(m), o.weighting[T])
        return x, u, c

    # ilqr.py:152-156 (objects.py:68-78,114-117) by finite differences
    def expand(self):
        o, n, m, T = self.o, self.n, self.m, self.T
        A, B = np.zeros((n, n, T), order="F"), np.zeros((n, m, T), order="F")
        Cxx, Cuu, Cxu = (np.zeros((n, n, T + 1), order="F"), np.zeros((m, m, T + 1), order="F"),
                         np.zeros((n, m, T + 1), order="F"))
        cx, cu = np.zeros((n, T + 1), order="F"), np.zeros((m, T + 1), order="F")
        upad = np.hstack((o.uref, np.zeros((m, 1))))
        for t in range(T + 1):
            x, u, ul, w = o.xref[:, t], upad[:, t], upad[:, t - 1], o.weighting[t]   # t = 0: u_last = the pad (ilqr/objects.py:72)
            if t < T:
                J = _central(lambda z: self.dyn(z[:n], z[n:]), np.concatenate((x, u)), 1e-6)
                A[:, :, t], B[:, :, t] = J[:, :n], J[:, n:]
            g = lambda z: _central(lambda y: self.cost(y[:n], y[n:], ul, w), z, 1e-5)[0]      # gradient of the scalar cost
            z0 = np.concatenate((x, u))
            grad = g(z0)
            H = _central(g, z0, 1e-4)
            H = 0.5 * (H + H.T)
            cx[:, t], cu[:, t] = grad[:n], grad[n:]
            Cxx[:, :, t], Cuu[:, :, t], Cxu[:, :, t] = H[:n, :n], H[n:, n:], H[:n, n:]
        return Cxx, cx, Cuu, cu, Cxu, A, B

    def run(self, nb_iter):
        from .objects import LinearControl, QuadraticStateActionValue, QuadraticStateValue
        o, n, m, T = self.o, self.n, self.m, self.T
        grow = lambda: (max(o.dlmbda * o.mult_lmbda, o.mult_lmbda),)
        trace = []
        for alpha in o.alphas:                                     # ilqr.py:139-147
            x, u, c = self.rollout(o.ctl, alpha)
            if np.all(x < 1e8):
                o.xref, o.uref, o.last_return = x, u, float(np.sum(c))
                break
        trace.append(o.last_return)
        for _ in range(nb_iter):
            Cxx, cx, Cuu, cu, Cxu, A, B = self.expand()
            passed, out = False, None
            while True:                                            # ilqr.py:162-173
                out, failed = core.backward_pass_flag(Cxx, cx, Cuu, cu, Cxu, A, B, o.lmbda, o.reg, n, m, T)
                if not failed:
                    passed = True
                    break
                o.dlmbda, = grow()
                o.lmbda = max(o.lmbda * o.dlmbda, o.min_lmbda)
                if o.lmbda > o.max_lmbda:
                    break
            Qxx, Qux, Quu, qx, qu, V, v, dV, K, kff, _div = out
            lc = LinearControl(n, m, T)
            lc.K, lc.kff = K, kff
            g_norm = np.mean(np.max(np.abs(kff) / (np.abs(o.uref) + 1.0), axis=1))
            if g_norm < o.tolgrad and o.lmbda < 1e-5:              # ilqr.py:176-180
                o.dlmbda = min(o.dlmbda / o.mult_lmbda, 1.0 / o.mult_lmbda)
                o.lmbda = o.lmbda * o.dlmbda * (o.lmbda > o.min_lmbda)
                o.status = 1
                break
            accepted = None
            if passed:
                for alpha in o.alphas:                             # ilqr.py:187-202
                    o.alpha = alpha
                    x, u, c = self.rollout(lc, alpha)
                    ret = float(np.sum(c))
                    dret = o.last_return - ret
                    expected = -1.0 * alpha * (dV[0] + alpha * dV[1])
                    if dret / expected >= o.min_imp:
                        accepted = (x, u, ret, dret)
                        break
            if accepted is not None:                               # ilqr.py:205-223
                o.dlmbda = min(o.dlmbda / o.mult_lmbda, 1.0 / o.mult_lmbda)
                o.lmbda = o.lmbda * o.dlmbda * (o.lmbda > o.min_lmbda)
                o.xref, o.uref, o.last_return = accepted[0], accepted[1], accepted[2]
                o.vfunc = QuadraticStateValue(n, T + 1)
                o.vfunc.V, o.vfunc.v = V, v
                o.qfunc = QuadraticStateActionValue(n, m, T)
                o.qfunc.Qxx, o.qfunc.Qux, o.qfunc.Quu, o.qfunc.qx, o.qfunc.qu = Qxx, Qux, Quu, qx, qu
                o.ctl = lc
                trace.append(o.last_return)
                if accepted[3] < o.tolfun:
                    o.status = 2
                    break
            else:                                                  # ilqr.py:224-231
                o.dlmbda, = grow()
                o.lmbda = max(o.lmbda * o.dlmbda, o.min_lmbda)
                if o.lmbda > o.max_lmbda:
                    o.status = 3
                    break
        return trace
