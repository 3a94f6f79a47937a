"""an env without a device model is expanded on the host and solved around the GPU backward pass
(trajopt_amd/ilqr/hostenv.py) - checked against the same env WITH its device model."""
import numpy as np
import pytest


class DuckPendulum:
    """The pendulum of /root/reference/trajopt/envs/pendulum/pendulum.py:58-125 as plain NumPy, with the reference's
    env interface (ilqr.py:23-31) and no struct(): iLQR has to take its host path."""

    class _Space:
        def __init__(self, high):
            self.high, self.shape = np.asarray(high, dtype=np.float64), (len(high),)

    def __init__(self, src):
        self.src = src
        self.dm_state, self.dm_act = 2, 1
        self.action_space, self.observation_space = self._Space(src.umax), self._Space(src.xmax)
        self.unwrapped = self
        self.slew_rate, self.uw = False, np.asarray(src.uw, dtype=np.float64)

    def dynamics(self, x, u):
        return self.src.dynamics(x, u)

    def cost(self, x, u, u_last, a):
        return self.src.cost(x, u, u_last, a)


@pytest.mark.gpu
def test_unknown_env_runs_through_the_host_path_and_matches_the_device_model():
    from trajopt_amd import _lib, envs
    from trajopt_amd.ilqr import BatchedILQR, iLQR
    T, iters = 40, 6
    kff = 1e-4 * np.random.default_rng(3).standard_normal((1, T))

    def solve(env):
        s = iLQR(env, T, np.array([np.pi * 0.9, 0.0]), action_penalty=None)
        s.ctl.kff = kff.copy()
        return s, s.run(nb_iter=iters)
    dev_env = envs.make("pendulum", dt=0.05, uw=1e-5)
    sd, trd = solve(dev_env)
    sh, trh = solve(DuckPendulum(envs.make("pendulum", dt=0.05, uw=1e-5)))
    assert len(trh) == len(trd) and len(trh) >= 3
    # finite-difference expansion on the host vs exact dual numbers on the device: the first iterations agree to FD
    # accuracy; later ones may accept another step size where the improvement ratio sits at the threshold
    np.testing.assert_allclose(trh[:5], trd[:5], rtol=1e-6)
    assert np.all(np.diff(trh) < 0) and abs(trh[-1] - trd[-1]) < 0.02 * abs(trd[-1])
    assert sh.ctl.K.shape == (1, 2, T) and sh.xref.shape == (2, T + 1) and sh.vfunc.V.shape == (2, 2, T + 1)
    with pytest.raises(_lib.TrajoptError):
        BatchedILQR(DuckPendulum(dev_env), T, np.zeros((4, 2)))
