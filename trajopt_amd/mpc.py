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

    def _plant(self, x, u, out=None):
        """xn = env.dynamics(x, u) for all problems; x (n, N), u (m, N) device tensors (PROBLEM_MINOR)."""
        import torch
        xn = torch.empty_like(x) if out is None else out
        e = self.solver._struct
        _lib.check(_lib.lib().trajopt_env_step_batched(C.byref(e), _lib.dptr(x), _lib.dptr(u), None, None,
                                                       C.c_int64(self.nb_problems), _lib.LAYOUT_PROBLEM_MINOR,
                                                       _lib.dptr(xn), None, _lib.stream_ptr()))
        return xn

    def run(self, nb_steps, process_noise=None, kff_inits=None, seed=1337, kff_rng="device", to_host=True, graph=False,
            warm_start=False):
        """-> dict(state (N, n, nb_steps + 1), action (N, m, nb_steps), cost (N, nb_steps): last accepted return of
        each horizon solve).  kff_inits: optional (nb_steps, N, m, horizon) initial feed-forward noise per solve.
        Without it every solve draws 1e-4 * randn as a fresh reference solver does (ilqr.py:70): kff_rng = "device"
        (default) from a torch generator ON THE GPU (seeded with `seed`; no host work, no upload - at 65 536 cartpoles x
        T = 500 the host draw is 262 MB per control step), kff_rng = "numpy" from np.random as the reference (to
        reproduce a reference run's stream).  to_host=False returns the device tensors (T-major, PROBLEM_MINOR).
        warm_start=True (SURVEY 8f.1's "shift warm-start"; the reference's scripts start every solve cold): from the second
        control step on a solve starts from the previous plan shifted by one step - kff_init[t] = uref_prev[t + 1], the last
        action repeated - instead of fresh noise; K = 0 and uref = 0 as in a fresh solver, so the initial rollout
        (ilqr.py:139-149) applies exactly that shifted plan.  Device work only (a slice copy of uref).
        graph=True: ONE control step - re-arm, initial rollout, nb_iter iterations, plant step: ~45 kernel launches, which
        at short horizons (the examples use 60) are most of its time - is captured once as a HIP graph (torch.cuda.CUDAGraph
        around the library's launches on the capture stream: nothing in that path allocates or synchronises) and replayed
        per step on fixed buffers; same kernels on the same data, same results (needs kff_rng = "device" or kff_inits).
        Measured (scripts/bench_mpc.py, cartpole, horizon 60, nb_iter 10): NO gain - 4096 systems 1.09 ms a control step
        eager, 1.17 as a graph; 512 systems 0.93 / 0.95; 65 536 systems 7.05 / 7.22.  The host enqueues the next launches
        while the GPU works, so the step is not bound by the host's launch calls but by the device side of ~45 dependent
        kernels of 5 - 50 us each (dispatch + drain per kernel), which a graph replays one by one all the same.  Kept as
        an option (and as the test that the whole step is capture-safe); what would shorten such a step is fewer,
        longer kernels."""
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
        assert not (warm_start and graph), "warm_start changes what a step starts from: use the eager loop"
        warm = None   # (N, m, horizon) logical view of the shifted previous plan
        if graph:
            assert kff_inits is not None or kff_rng == "device", "graph=True draws the kff noise on the device"
            x_cur = state[0].clone()
            u_out = torch.empty((m, N), dtype=torch.float64, device=dev)
            c_out = torch.empty((N,), dtype=torch.float64, device=dev)
            noise = torch.zeros((n, N), dtype=torch.float64, device=dev) if (process_noise is not None and gen is None) else None
            kbuf = None
            if kff_inits is not None:
                kbuf = torch.empty((N, m, self.horizon), dtype=torch.float64, device=dev)
            s._reserve_trace(self.nb_iter + 2)   # (no reallocation inside the captured region)
            g = torch.cuda.CUDAGraph()
            for rg in (kgen, gen):
                if rg is not None:
                    g.register_generator_state(rg)
            with torch.cuda.graph(g):
                s.reset(x_cur.t(), kbuf, generator=kgen)
                s.init_rollout()
                s.iterate(self.nb_iter)
                u_out.copy_(s.uref[0])
                c_out.copy_(s.last_return)
                xn = self._plant(x_cur, u_out)
                if gen is not None:
                    xn = xn + chol @ torch.randn((n, N), dtype=torch.float64, device=dev, generator=gen)
                elif noise is not None:
                    xn = xn + noise
                x_cur.copy_(xn)
            for t in range(nb_steps):
                if kbuf is not None:
                    kbuf.copy_(torch.as_tensor(np.asarray(kff_inits[t], dtype=np.float64).reshape(N, m, self.horizon)))
                if noise is not None:
                    noise.copy_(process_noise[:, :, t].t())
                g.replay()
                action[t].copy_(u_out)
                cost[t].copy_(c_out)
                state[t + 1].copy_(x_cur)
            nb_steps = 0   # (the eager loop below has nothing left to do)
        for t in range(nb_steps):
            if warm is not None:
                s.reset(state[t].t(), warm, generator=None)
            else:
                s.reset(state[t].t(), None if kff_inits is None else kff_inits[t], generator=kgen)
            s.init_rollout()
            s.iterate(self.nb_iter)
            action[t] = s.uref[0]
            cost[t] = s.last_return
            if warm_start:
                # uref is (horizon, m, N) on the device: the plan from its second action on, the last one twice
                shifted = torch.cat((s.uref[1:], s.uref[-1:]), dim=0)
                warm = shifted.permute(2, 1, 0)
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
