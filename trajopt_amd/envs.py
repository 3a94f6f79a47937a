"""Host-side mirror of the reference's env layer for the envs the hot path names (row A8).

The reference's envs are gym.Env subclasses written in autograd.numpy
(/root/reference/trajopt/envs/{pendulum/pendulum.py, cartpole/cartpole.py,
double_cartpole/double_cartpole.py, lqr/lqr_v0.py}) and are consumed by iLQR only through
`env.unwrapped.dynamics / .cost`, `env.action_space.high / .shape`, `env.observation_space.shape`
and the writable `env.unwrapped.slew_rate / .uw` (/root/reference/trajopt/ilqr/ilqr.py:23-36).
Here an env is a parameter bag with the same attribute names (dm_state, dm_act, dt, g, gw, uw, umax,
xmax, x0, sigma0, sigma, slew_rate, periodic) whose `struct()` is the POD the HIP kernels take
(include/trajopt_env.h); `dynamics` and `cost` evaluate ON THE DEVICE through
trajopt_env_step_batched - there is no host implementation of the physics in this package.
gym is not imported (absent in the build image); `make()` understands the reference's ids.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (ENV_CARTPOLE, ENV_DOUBLE_CARTPOLE, ENV_DOUBLE_PENDULUM, ENV_LQR, ENV_PENDULUM, ENV_QUAD_PENDULUM, ENV_QUADROTOR, ENV_ROBOT, MAX_ACT, MAX_STATE,
                   EnvStruct)


class Box:
    """The two attributes of gym.spaces.Box the solvers read."""

    def __init__(self, low, high):
        self.low = np.atleast_1d(np.asarray(low, dtype=np.float64))
        self.high = np.atleast_1d(np.asarray(high, dtype=np.float64))
        self.shape = self.high.shape


class Env:
    model = None
    phys = ()
    angles = ()          # indices wrapped by the cost when `periodic` (e.g. cartpole.py:121-122)
    features = None      # [(kind, state)], kind 0 = id, 1 = cos, 2 = sin: the ...WithCartesianCost classes; None = identity
    cost_times_dt = False

    def __init__(self):
        self.slew_rate = False
        self.periodic = False
        self.state = None
        self.np_random = None
        self.seed()

    # -- gym surface ---------------------------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    @property
    def observation_space(self):
        return Box(-np.asarray(self.xmax), np.asarray(self.xmax))

    @property
    def action_space(self):
        u = np.broadcast_to(np.asarray(self.umax, dtype=np.float64), (self.dm_act,))
        return Box(-u, u)

    @property
    def xlim(self):
        return self.xmax

    @property
    def ulim(self):
        return self.umax

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)  # gym.utils.seeding.np_random of the reference's era
        return [seed]

    def init(self):
        return self.x0, self.sigma0

    def reset(self):
        self.state = self.np_random.multivariate_normal(mean=self.x0, cov=self.sigma0)
        return self.state

    def noise(self, x=None, u=None):
        return self.sigma

    def step(self, u):
        _sigma = self.noise(self.state, u)
        self.state = self.dynamics(self.state, u)
        self.state = self.np_random.multivariate_normal(mean=self.state, cov=_sigma)
        return self.state, [], False, {}

    # -- device struct -------------------------------------------------------------------------
    def struct(self):
        """include/trajopt_env.h::trajopt_env_t from the CURRENT attribute values (the reference's
        scripts mutate env.unwrapped.dt / .uw / .slew_rate after construction)."""
        e = EnvStruct()
        n, m = int(self.dm_state), int(self.dm_act)
        if n > MAX_STATE or m > MAX_ACT:
            raise _lib.TrajoptError("env dims (%d, %d) exceed (%d, %d)" % (n, m, MAX_STATE, MAX_ACT))
        e.model, e.dm_state, e.dm_act = int(self.model), n, m
        e.slew_rate = int(bool(self.slew_rate))
        e.periodic = sum(1 << i for i in self.angles) if self.periodic else 0
        e.cost_times_dt = int(bool(self.cost_times_dt))
        e.dt = float(self.dt)

        def fill(field, vals, cnt):
            vals = np.broadcast_to(np.asarray(vals, dtype=np.float64), (cnt,))
            for i in range(cnt):
                field[i] = float(vals[i])

        fill(e.umax, self.umax, m)
        fill(e.xmax, self.xmax, n)
        nf = n
        if self.features is not None:
            nf = len(self.features)
            if nf > MAX_STATE:
                raise _lib.TrajoptError("%d cost features exceed %d" % (nf, MAX_STATE))
            e.nb_features = nf
            for f, (kind, i) in enumerate(self.features):
                e.features[f] = (int(kind) << 8) | int(i)
        fill(e.goal, self.g, nf)
        fill(e.gw, self.gw, nf)
        fill(e.uw, self.uw, m)
        for i, v in enumerate(self.phys):
            e.phys[i] = float(v)
        self._fill_linear(e)
        return e

    def _fill_linear(self, e):
        pass

    # -- single evaluations, on the device -----------------------------------------------------
    def _step(self, x, u, u_last, a, want_cost):
        import torch
        _lib.require_gpu()
        dev = torch.device("cuda", torch.cuda.current_device())
        n, m = self.dm_state, self.dm_act
        xd = torch.as_tensor(np.asarray(x, dtype=np.float64).reshape(n), device=dev)
        ud = torch.as_tensor(np.asarray(u, dtype=np.float64).reshape(m), device=dev)
        uld = None if u_last is None else torch.as_tensor(np.asarray(u_last, dtype=np.float64).reshape(m), device=dev)
        ad = None if a is None else torch.tensor([float(a)], dtype=torch.float64, device=dev)
        out = torch.empty(1 if want_cost else n, dtype=torch.float64, device=dev)
        e = self.struct()
        _lib.check(_lib.lib().trajopt_env_step_batched(
            C.byref(e), _lib.dptr(xd), _lib.dptr(ud), _lib.dptr(uld), _lib.dptr(ad), C.c_int64(1),
            _lib.LAYOUT_PROBLEM_MINOR, None if want_cost else _lib.dptr(out), _lib.dptr(out) if want_cost else None,
            _lib.stream_ptr()))
        r = out.cpu().numpy()
        return float(r[0]) if want_cost else r

    def dynamics(self, x, u):
        """xn = clip(RK4(x, clip(u)))  (e.g. cartpole.py:62-97)."""
        return self._step(x, u, None, None, False)

    def cost(self, x, u, u_last, a):
        """u' diag(uw) u (or the slew form) + a (y - g)' diag(gw) (y - g)  (e.g. cartpole.py:112-127)."""
        return self._step(x, u, u_last, a, True)


class Pendulum(Env):
    """/root/reference/trajopt/envs/pendulum/pendulum.py:15-149"""
    model, angles = ENV_PENDULUM, (0,)
    phys = (9.81, 1., 1., 1e-3)  # g, m, l, k  (pendulum.py:61)

    def __init__(self):
        self.dm_state, self.dm_act, self.dt = 2, 1, 0.01
        self.sigma = 1e-4 * np.eye(2)
        self.g, self.gw = np.array([0., 0.]), np.array([1e0, 1e-1])
        self.xmax = np.array([np.inf, np.inf])
        self.uw, self.umax = 1e-3 * np.ones(1), 10. * np.ones(1)
        self.x0, self.sigma0 = np.array([np.pi, 0.]), 1e-4 * np.eye(2)
        super().__init__()

    def reset(self):  # pendulum.py:143-149
        low, high = np.array([np.pi - np.pi / 10., -0.75]), np.array([np.pi + np.pi / 10., 0.75])
        x0 = self.np_random.uniform(low=low, high=high)
        self.state = np.hstack((((x0[0] + np.pi) % (2. * np.pi)) - np.pi, x0[1]))
        return self.state


class Cartpole(Env):
    """/root/reference/trajopt/envs/cartpole/cartpole.py:15-147"""
    model, angles = ENV_CARTPOLE, (1,)
    phys = (9.81, 0.37, 0.127, 0.3365, 0.005)  # g, Mc, Mp, l, fr  (cartpole.py:67-72)

    def __init__(self):
        self.dm_state, self.dm_act, self.dt = 4, 1, 0.01
        self.sigma = 1e-8 * np.eye(4)
        self.g, self.gw = np.zeros(4), np.array([1e1, 1e4, 1e0, 1e0])
        self.xmax = np.array([10., np.inf, np.inf, np.inf])
        self.uw, self.umax = 1e-5 * np.ones(1), 10. * np.ones(1)
        self.x0, self.sigma0 = np.array([0., np.pi, 0., 0.]), 1e-4 * np.eye(4)
        super().__init__()


class DoubleCartpole(Env):
    """/root/reference/trajopt/envs/double_cartpole/double_cartpole.py:15-205"""
    model, angles = ENV_DOUBLE_CARTPOLE, (1, 2)
    phys = (9.81, 0.37, 0.127, 0.127, 0.3365, 0.3365)  # g, Mc, Mp1, Mp2, L1, L2  (double_cartpole.py:71-77)

    def __init__(self):
        self.dm_state, self.dm_act, self.dt = 6, 1, 0.01
        self.sigma = 1e-8 * np.eye(6)
        self.g, self.gw = np.zeros(6), np.array([1e1, 1e4, 1e4, 1e0, 1e0, 1e0])
        self.xmax = np.array([10.] + [np.inf] * 5)
        self.uw, self.umax = 1e-5 * np.ones(1), 5. * np.ones(1)
        self.x0, self.sigma0 = np.array([0., np.pi, np.pi, 0., 0., 0.]), 1e-4 * np.eye(6)
        super().__init__()


class DoublePendulum(Env):
    """/root/reference/trajopt/envs/double_pendulum/double_pendulum.py:15-172 (a further env model
    behind the same interface; two torque-driven links, dx = 4, du = 2)."""
    model, angles = ENV_DOUBLE_PENDULUM, (0,)
    phys = (9.81, 1., 1., 1., 1., 0.025, 0.025)  # g, m1, m2, l1, l2, k1, k2  (double_pendulum.py:66-70)

    def __init__(self):
        self.dm_state, self.dm_act, self.dt = 4, 2, 0.01
        self.sigma = 1e-8 * np.eye(4)
        self.g, self.gw = np.zeros(4), np.array([1e4, 1e4, 1e0, 1e0])
        self.xmax = np.full(4, np.inf)
        self.uw, self.umax = 1e-5 * np.ones(2), 10. * np.ones(2)
        self.x0, self.sigma0 = np.array([np.pi, 0., 0., 0.]), 1e-4 * np.eye(4)
        super().__init__()


class QuadPendulum(Env):
    """/root/reference/trajopt/envs/quad_pendulum/quad_pendulum.py:14-303: four torque-driven links, dx = 8, du = 4."""
    model, angles = ENV_QUAD_PENDULUM, (0, 1, 2, 3)
    phys = (9.81,) + (1.,) * 4 + (1.,) * 4 + (0.025,) * 4   # G, masses, lengths, friction (quad_pendulum.py:68-72)

    def __init__(self):
        self.dm_state, self.dm_act, self.dt = 8, 4, 0.01
        self.sigma = 1e-8 * np.eye(8)
        self.g, self.gw = np.zeros(8), np.array([1e4] * 4 + [1e0] * 4)
        self.xmax = np.full(8, np.inf)
        self.uw, self.umax = 1e-5 * np.ones(4), 25. * np.ones(4)
        self.x0, self.sigma0 = np.array([np.pi] + [0.] * 7), 1e-4 * np.eye(8)
        super().__init__()


class QuadPendulumWithCartesianCost(QuadPendulum):
    """/root/reference/trajopt/envs/quad_pendulum/quad_pendulum.py:281-303: cost on [cos, sin of the four angles, velocities]
    (12 features: exactly the capacity of trajopt_env_t::goal / gw)."""
    features = ((1, 0), (2, 0), (1, 1), (2, 1), (1, 2), (2, 2), (1, 3), (2, 3), (0, 4), (0, 5), (0, 6), (0, 7))

    def __init__(self):
        super().__init__()
        self.g, self.gw = np.array([1., 0.] * 4 + [0.] * 4), np.array([1e4] * 8 + [1e0] * 4)


class PendulumWithCartesianCost(Pendulum):
    """/root/reference/trajopt/envs/pendulum/pendulum.py:152-162: cost on [cos th, sin th, dth]."""
    features = ((1, 0), (2, 0), (0, 1))

    def __init__(self):
        super().__init__()
        self.g, self.gw = np.array([1., 0., 0.]), np.array([1e4, 1e4, 1e0])


class CartpoleWithCartesianCost(Cartpole):
    """/root/reference/trajopt/envs/cartpole/cartpole.py:150-166: cost on [x, cos th, sin th, dx, dth]."""
    features = ((0, 0), (1, 1), (2, 1), (0, 2), (0, 3))

    def __init__(self):
        super().__init__()
        self.g, self.gw = np.array([0., 1., 0., 0., 0.]), np.array([1e1, 1e4, 1e4, 1e0, 1e0])


class DoubleCartpoleWithCartesianCost(DoubleCartpole):
    """/root/reference/trajopt/envs/double_cartpole/double_cartpole.py:210-230.  Its feature map reads
    [x0, cos x0, sin x0, cos x1, sin x1, x2, x3, x4] - one index short of its own comment, x5 unused - kept as written."""
    features = ((0, 0), (1, 0), (2, 0), (1, 1), (2, 1), (0, 2), (0, 3), (0, 4))

    def __init__(self):
        super().__init__()
        self.g = np.array([0., 1., 0., 1., 0., 0., 0., 0.])
        self.gw = np.array([1e1, 1e4, 1e4, 1e4, 1e4, 1e0, 1e0, 1e0])


class DoublePendulumWithCartesianCost(DoublePendulum):
    """/root/reference/trajopt/envs/double_pendulum/double_pendulum.py:178-194."""
    features = ((1, 0), (2, 0), (1, 1), (2, 1), (0, 2), (0, 3))

    def __init__(self):
        super().__init__()
        self.g, self.gw = np.array([1., 0., 1., 0., 0., 0.]), np.array([1e4, 1e4, 1e4, 1e4, 1e0, 1e0])


class LQRv0(Env):
    """/root/reference/trajopt/envs/lqr/lqr_v0.py:8-114: xn = A x + B clip(u) + c, cost scaled by dt.
    A, B, c, g, gw, uw may be replaced (any dx <= 12, du <= 4) for the time-invariant LQ known-answer cases."""
    model, cost_times_dt = ENV_LQR, True

    def __init__(self, A=None, B=None, c=None, g=None, gw=None, uw=None, x0=None, dt=0.1):
        self.A = np.array([[1.1, 0.], [1.0, 1.0]]) if A is None else np.asarray(A, dtype=np.float64)
        self.B = np.array([[1.], [0.]]) if B is None else np.asarray(B, dtype=np.float64)
        self.dm_state, self.dm_act = self.B.shape
        self.dt = dt
        self.g = np.full(self.dm_state, 10.) if g is None else np.asarray(g, dtype=np.float64)
        self.c = -self.A @ self.g if c is None else np.asarray(c, dtype=np.float64)
        self.gw = np.full(self.dm_state, 1e1) if gw is None else np.asarray(gw, dtype=np.float64)
        self.uw = np.full(self.dm_act, 1e0) if uw is None else np.asarray(uw, dtype=np.float64)
        self.umax, self.xmax = np.inf, np.full(self.dm_state, np.inf)
        self.x0 = np.full(self.dm_state, 5.) if x0 is None else np.asarray(x0, dtype=np.float64)
        self.sigma, self.sigma0 = 1e-8 * np.eye(self.dm_state), 1e-2 * np.eye(self.dm_state)
        super().__init__()

    def _fill_linear(self, e):
        n, m = self.dm_state, self.dm_act
        for k, v in enumerate(np.asarray(self.A).flatten(order="F")):
            e.lin_A[k] = float(v)
        for k, v in enumerate(np.asarray(self.B).flatten(order="F")):
            e.lin_B[k] = float(v)
        for k, v in enumerate(np.asarray(self.c).reshape(n)):
            e.lin_c[k] = float(v)


class LQRv1(LQRv0):
    """/root/reference/trajopt/envs/lqr/lqr_v1.py:8-109: Euler-discretised mass-spring-damper (the deterministic
    `dist=None` dynamics; the sampled-parameter branch :60-65 belongs to rGPS, out of scope)."""

    def __init__(self, dt=0.01):
        m, k, d = 1., 1e-2, 1e-1
        super().__init__(A=np.array([[0., 1.], [-k / m, -d / m]]) * dt + np.eye(2), B=np.array([[0.], [1. / m]]) * dt,
                         c=np.zeros(2), g=np.array([1., 0.]), gw=np.array([1e2, 1e0]), uw=np.array([1e-3]),
                         x0=np.zeros(2), dt=dt)
        self.sigma0 = 1e-4 * np.eye(2)


class LQRv2(LQRv0):
    """/root/reference/trajopt/envs/lqr/lqr_v2.py:8-106 (deterministic dynamics, as LQRv1)."""

    def __init__(self, dt=0.01):
        super().__init__(A=np.array([[1.0, 0.], [0.1, 1.1]]), B=np.array([[0.05], [0.]]), c=np.zeros(2),
                         g=np.array([1., 0.]), gw=np.array([1e1, 1e0]), uw=np.array([1e-3]), x0=np.full(2, 5.), dt=dt)


class Robot(Env):
    """/root/reference/trajopt/envs/robot/robot.py:8-102: car-like robot x = [x, y, th, v], u = [a, phi]; cost scaled
    by dt; its RK4 steps by -dt (robot.py:62-67), kept."""
    model, cost_times_dt = ENV_ROBOT, True
    phys = (0.1,)  # car length (robot.py:17)

    def __init__(self):
        self.dm_state, self.dm_act, self.dt = 4, 2, 0.025
        self.sigma = 1e-8 * np.eye(4)
        self.g, self.gw = np.zeros(4), np.array([1e1, 1e1, 1., 1.])
        self.xmax = np.full(4, np.inf)
        self.uw, self.umax = 1e-3 * np.ones(2), np.full(2, np.inf)
        self.x0, self.sigma0 = np.array([5., 5., 0., 0.]), 1e-4 * np.eye(4)
        super().__init__()


class Quadrotor(Env):
    """Build-defined 12-state rigid body with 4 rotor thrusts (BASELINE config C4).  NOT in the reference:
    its nearest env is QuadPendulum (dx=8, du=4).  x = [p(3), roll/pitch/yaw, v(3), body rates], u = thrusts."""
    model = ENV_QUADROTOR
    phys = (9.81, 0.5, 0.25, 3.9e-3, 3.9e-3, 7.5e-3, 1.6e-2)  # g, mass, arm, Ixx, Iyy, Izz, kM

    def __init__(self):
        self.dm_state, self.dm_act, self.dt = 12, 4, 0.01
        self.sigma = 1e-8 * np.eye(12)
        self.g = np.array([1., 1., 1.] + [0.] * 9)
        self.gw = np.array([1e1] * 3 + [1e0] * 3 + [1e0] * 3 + [1e-1] * 3)
        self.xmax = np.full(12, np.inf)
        self.uw, self.umax = 1e-3 * np.ones(4), 5. * np.ones(4)
        self.x0, self.sigma0 = np.zeros(12), 1e-4 * np.eye(12)
        super().__init__()


_REGISTRY = {
    # the reference's gym ids (/root/reference/trajopt/__init__.py:3-97) for the envs in scope
    "Pendulum-TO-v0": Pendulum, "Cartpole-TO-v0": Cartpole, "DoubleCartpole-TO-v0": DoubleCartpole,
    "LQR-TO-v0": LQRv0, "DoublePendulum-TO-v0": DoublePendulum, "double_pendulum": DoublePendulum,
    "Pendulum-TO-v1": PendulumWithCartesianCost, "Cartpole-TO-v1": CartpoleWithCartesianCost,
    "DoubleCartpole-TO-v1": DoubleCartpoleWithCartesianCost, "DoublePendulum-TO-v1": DoublePendulumWithCartesianCost,
    "QuadPendulum-TO-v1": QuadPendulumWithCartesianCost, "quad_pendulum_cartesian": QuadPendulumWithCartesianCost,
    "pendulum_cartesian": PendulumWithCartesianCost, "cartpole_cartesian": CartpoleWithCartesianCost,
    "double_cartpole_cartesian": DoubleCartpoleWithCartesianCost,
    "double_pendulum_cartesian": DoublePendulumWithCartesianCost,
    "QuadPendulum-TO-v0": QuadPendulum, "quad_pendulum": QuadPendulum,
    "LQR-TO-v1": LQRv1, "LQR-TO-v2": LQRv2, "Robot-TO-v0": Robot, "lqr_v1": LQRv1, "lqr_v2": LQRv2, "robot": Robot,
    "pendulum": Pendulum, "cartpole": Cartpole, "double_cartpole": DoubleCartpole, "lqr": LQRv0,
    "quadrotor": Quadrotor, "Quadrotor-TO-v0": Quadrotor,
}


def make(name, dt=None, uw=None, slew_rate=None, periodic=None, cartesian=False, **kw):
    """gym.make stand-in: `make('Cartpole-TO-v0')`; keyword overrides are set as attributes afterwards,
    the way the reference's scripts do (examples/ilqr/mpc/pendulum.py:12-13)."""
    if cartesian:
        name = name + "_cartesian"
    if name not in _REGISTRY:
        raise KeyError("unknown env %r (in scope: %s)" % (name, sorted(_REGISTRY)))
    env = _REGISTRY[name](**kw)
    if dt is not None:
        env.dt = float(dt)
    if uw is not None:
        env.uw = np.asarray(uw, dtype=np.float64) * np.ones(env.dm_act)
    if slew_rate is not None:
        env.slew_rate = bool(slew_rate)
    if periodic is not None:
        env.periodic = bool(periodic)
    return env
