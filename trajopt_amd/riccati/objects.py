"""Containers of the reference's Riccati solver (/root/reference/trajopt/riccati/objects.py): the
"no-1/2, affine-c" convention.  taylor_expansion runs the HIP quadratisation kernel with convention 1.
"""
import numpy as np

from ..ilqr.objects import QuadraticCost, QuadraticStateValue, quadratize  # noqa: F401  (same containers)


class LinearDynamics:  # riccati/objects.py:68-87
    def __init__(self, dm_state, dm_act, nb_steps):
        self.dm_state, self.dm_act, self.nb_steps = dm_state, dm_act, nb_steps
        self.A = np.zeros((dm_state, dm_state, nb_steps))
        self.B = np.zeros((dm_state, dm_act, nb_steps))
        self.c = np.zeros((dm_state, nb_steps))

    @property
    def params(self):
        return self.A, self.B, self.c

    @params.setter
    def params(self, values):
        self.A, self.B, self.c = values

    def sample(self, x, u):
        pass


class AnalyticalLinearDynamics(LinearDynamics):  # riccati/objects.py:90-108
    def __init__(self, f_dyn, dm_state, dm_act, nb_steps):
        super().__init__(dm_state, dm_act, nb_steps)
        self.f = f_dyn
        self.env = getattr(f_dyn, "__self__", None)

    def evalf(self, x, u):
        return self.f(x, u)

    def taylor_expansion(self, x, u):
        q = quadratize(self.env, self.nb_steps, x, u, np.ones(self.nb_steps + 1), convention=1)
        self.A, self.B, self.c = q["A"], q["B"], q["c"]


class AnalyticalQuadraticCost(QuadraticCost):  # riccati/objects.py:38-65
    def __init__(self, f, dm_state, dm_act, nb_steps):
        super().__init__(dm_state, dm_act, nb_steps)
        self.f = f
        self.env = getattr(f, "__self__", None)

    def evalf(self, x, u, u_last, a):
        return self.f(x, u, u_last, a)

    def taylor_expansion(self, x, u, a):
        q = quadratize(self.env, self.nb_steps - 1, x, u, a, convention=1)
        self.Cxx, self.cx, self.Cuu, self.cu, self.Cxu = q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"]


class LinearControl:  # riccati/objects.py:111-129 - absolute law u = kff + K x
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

    def action(self, x, t):
        return self.kff[..., t] + self.K[..., t] @ x
