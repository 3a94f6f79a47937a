"""Batched receding-horizon MPC with iLQR (first "next" row).

The reference's MPC scripts (/root/reference/examples/ilqr/mpc/pendulum.py:18-31, cartpole.py:18-32,
doublecartpole.py:18-32) do, for ONE system and every control step t:

    solver = iLQR(env, init_state=state[:, t], nb_steps=horizon, action_penalty=...)   # a fresh solver
    solver.run(nb_iter=10)
    action[:, t] = solver.uref[:, 0]
    state[:, t + 1] = env.step(action[:, t])                                           # noisy plant

and pendulum_parallel.py:40-45 fans 50 such loops over OS processes.  Here N loops advance together: one
BatchedILQR workspace is re-armed at every step (no allocation inside the loop) and the plant is stepped on the
device (trajopt_env_step_batched).  Everything numerical runs in the HIP library.
"""
import ctypes as C

import numpy as np

from . import _lib
from .ilqr import BatchedILQR


class BatchedMPC:
    """N independent MPC loops of one env model on the current GPU.

    kwargs are those of iLQR / BatchedILQR (horizon solver).  `process_noise`: None = deterministic plant,
    "env" = x_{t+1} ~ N(dynamics(x_t, u_t), env.sigma) drawn from a torch generator (the reference draws from the
    env's NumPy RNG, so noisy runs agree in distribution only), or an explicit (N, dm_state, nb_steps) array that
    is added to the plant's output (reproducible)."""

    def __init__(self, env, horizon, init_states, nb_iter=10, **ilqr_kwargs):
        init_states = np.asarray(init_states, dtype=np.float64)
        self.env, self.horizon, self.nb_iter = env, int(horizon), int(nb_iter)
        self.nb_problems, self.dm_state = init_states.shape
        self.solver = BatchedILQR(env, horizon, init_states, **ilqr_kwargs)
        self.dm_act = self.solver.dm_act
        self._x0 = init_states

    def _plant(self, x, u):
        """xn = env.dynamics(x, u) for all problems; x (n, N), u (m, N) device tensors (PROBLEM_MINOR)."""
        import torch
        xn = torch.empty_like(x)
        e = self.solver._struct
        _lib.check(_lib.lib().trajopt_env_step_batched(C.byref(e), _lib.dptr(x), _lib.dptr(u), None, None,
                                                       C.c_int64(self.nb_problems), _lib.LAYOUT_PROBLEM_MINOR,
                                                       _lib.dptr(xn), None, _lib.stream_ptr()))
        return xn

    def run(self, nb_steps, process_noise=None, kff_inits=None, seed=1337, kff_rng="device", to_host=True):
        """-> dict(state (N, n, nb_steps + 1), action (N, m, nb_steps), cost (N, nb_steps): last accepted return of
        each horizon solve).  kff_inits: optional (nb_steps, N, m, horizon) initial feed-forward noise per solve.
        Without it every solve draws 1e-4 * randn as a fresh reference solver does (ilqr.py:70): kff_rng = "device"
        (default) from a torch generator ON THE GPU (seeded with `seed`; no host work, no upload - at 65 536 cartpoles x
        T = 500 the host draw is 262 MB per control step), kff_rng = "numpy" from np.random as the reference (to
        reproduce a reference run's stream).  to_host=False returns the device tensors (T-major, PROBLEM_MINOR)."""
        import torch
        s = self.solver
        N, n, m = self.nb_problems, self.dm_state, self.dm_act
        dev = s.device
        state = torch.empty((nb_steps + 1, n, N), dtype=torch.float64, device=dev)
        action = torch.empty((nb_steps, m, N), dtype=torch.float64, device=dev)
        cost = torch.empty((nb_steps, N), dtype=torch.float64, device=dev)
        state[0] = torch.as_tensor(self._x0, device=dev).t()
        assert kff_rng in ("device", "numpy")
        kgen = None
        if kff_inits is None and kff_rng == "device":
            kgen = torch.Generator(device=dev)
            kgen.manual_seed(seed + 977)
        gen = None
        chol = None
        if isinstance(process_noise, str):
            assert process_noise == "env"
            gen = torch.Generator(device=dev)
            gen.manual_seed(seed)
            chol = torch.as_tensor(np.linalg.cholesky(np.asarray(self.env.unwrapped.sigma, dtype=np.float64)), device=dev)
        elif process_noise is not None:
            process_noise = torch.as_tensor(np.asarray(process_noise, dtype=np.float64), device=dev)  # (N, n, steps)
        for t in range(nb_steps):
            s.reset(state[t].t(), None if kff_inits is None else kff_inits[t], generator=kgen)
            s.init_rollout()
            s.iterate(self.nb_iter)
            action[t] = s.uref[0]
            cost[t] = s.last_return
            xn = self._plant(state[t].contiguous(), action[t].contiguous())
            if gen is not None:
                xn = xn + chol @ torch.randn((n, N), dtype=torch.float64, device=dev, generator=gen)
            elif process_noise is not None:
                xn = xn + process_noise[:, :, t].t()
            state[t + 1] = xn
        torch.cuda.synchronize(dev)
        if not to_host:
            return dict(state=state, action=action, cost=cost)
        return dict(state=state.permute(2, 1, 0).cpu().numpy(), action=action.permute(2, 1, 0).cpu().numpy(),
                    cost=cost.t().cpu().numpy())
