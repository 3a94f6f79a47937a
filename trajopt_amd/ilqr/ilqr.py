"""iLQR solver drivers.

`iLQR`        - the reference's class (/root/reference/trajopt/ilqr/ilqr.py:10-237): same constructor
                kwargs, methods, attributes and NumPy shapes, one problem.
`BatchedILQR` - the same algorithm for N independent problems resident on one GPU: every per-problem
                quantity of iLQR.run (xref, uref, ctl, lmbda, dlmbda, last_return, trace ...) is a device
                tensor in PROBLEM_MINOR layout and an iteration is a fixed sequence of HIP kernels with
                no host synchronisation (trajopt_amd/csrc/driver.hip).
Both run the same kernels; `iLQR` is `BatchedILQR` with N = 1 plus host copies.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import (ILQROptions, ILQRWorkspace, LAYOUT_PROBLEM_MAJOR, LAYOUT_PROBLEM_MINOR, check, dptr, lib,
                    stream_ptr)
from ..layout import empty as lempty, from_device, to_device
from . import core
from .objects import (AnalyticalLinearDynamics, AnalyticalQuadraticCost, LinearControl, QuadraticStateActionValue,
                      QuadraticStateValue)

DEFAULT_ALPHAS = np.power(10., np.linspace(0, -3, 11))  # ilqr.py:15

STATUS_RUNNING, STATUS_TOLGRAD, STATUS_TOLFUN, STATUS_MAX_LMBDA = 0, 1, 2, 3


def make_weighting(nb_steps, activation):
    """ilqr.py:73-83 (including its `"mult" and "shift" in activation` test, which only looks for "shift")."""
    if activation is None:
        return np.ones((nb_steps + 1,))
    if "mult" and "shift" in activation:
        t = np.linspace(0, nb_steps, nb_steps + 1)
        return 1. / (1. + np.exp(-activation['mult'] * (t - activation['shift'])))
    if "discount" in activation:
        w = np.ones((nb_steps + 1,))
        w[1:] = np.cumprod(activation["discount"] * np.ones((nb_steps,)))
        return w
    raise NotImplementedError


class BatchedILQR:
    """N problems of one env model on the current GPU.

    init_state: (N, dm_state) array (or torch tensor).  kff_init: (N, dm_act, nb_steps) initial feed-forward
    (default: 1e-4 * randn from np.random, as ilqr.py:70).  Remaining kwargs as the reference's iLQR.
    argmin_alpha=True picks the lowest-return accepted alpha instead of the reference's first-accept order.
    keep_value=True also maintains vfunc/qfunc (V, v, Qxx, Qux, Quu, qx, qu) of the accepted iterate.
    Device tensors are exposed as attributes in PROBLEM_MINOR shape (T, cols, rows, N), except the "gains
    group" (K, kff, vfunc/qfunc arrays and the quadratisation scratch), whose layout `gains_layout` is chosen for
    the backward kernel: PROBLEM_MINOR for dm_state <= 6 (lane-per-problem kernel), PROBLEM_MAJOR (N, T, cols, rows)
    above (matrix-core kernel).  `.host(name)` returns the logical (N, rows, cols, T) NumPy array either way.
    """

    GAINS_GROUP = ("K", "kff", "V", "v", "Qxx", "Qux", "Quu", "qx", "qu")

    def __init__(self, env, nb_steps, init_state, kff_init=None, activation=None, slew_rate=False,
                 action_penalty=False, alphas=DEFAULT_ALPHAS, lmbda=1., dlmbda=1., min_lmbda=1e-6, max_lmbda=1e6,
                 mult_lmbda=1.6, tolfun=1e-6, tolgrad=1e-4, min_imp=0., reg=1, argmin_alpha=False,
                 keep_value=False, trace_cap=None, device=None, apply_env_kwargs=True, gains_layout=None,
                 staged_line_search=-1, no_fusion=False):
        import torch
        _lib.require_gpu()
        self.env = env
        u = env.unwrapped if hasattr(env, "unwrapped") else env
        if not hasattr(u, "struct"):
            raise _lib.TrajoptError(
                "BatchedILQR needs an env with a device model (trajopt_amd.envs: %s has no struct()); for an arbitrary "
                "Python env use the one-problem class iLQR, which differentiates it on the host and runs the backward "
                "pass on the GPU (trajopt_amd/ilqr/hostenv.py)" % type(u).__name__)
        if apply_env_kwargs:  # ilqr.py:34-36
            u.slew_rate = slew_rate
            if action_penalty is not None:
                u.uw = action_penalty * np.ones((u.dm_act,))
        self.dm_state, self.dm_act, self.nb_steps = int(u.dm_state), int(u.dm_act), int(nb_steps)
        n, m, T = self.dm_state, self.dm_act, self.nb_steps
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        dev = self.device
        if torch.is_tensor(init_state):
            x0 = init_state.detach().to(device=dev, dtype=torch.float64)
        else:
            x0 = torch.as_tensor(np.asarray(init_state, dtype=np.float64), device=dev)
        if x0.dim() == 1:
            x0 = x0.reshape(1, n)
        assert x0.shape[1] == n, "init_state must be (N, dm_state)"
        self.nb_problems = N = int(x0.shape[0])
        self.alphas = np.asarray(alphas, dtype=np.float64)
        assert 1 <= self.alphas.size <= _lib.MAX_ALPHAS
        self.weighting_host = make_weighting(T, activation)
        self._struct = u.struct()

        self.gains_layout = GL = (LAYOUT_PROBLEM_MINOR if n <= 6 else LAYOUT_PROBLEM_MAJOR) \
            if gains_layout is None else int(gains_layout)
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        z = lambda *s: torch.zeros(s, **f64)
        zg = lambda *logical: lempty((N,) + logical, GL, dev, zero=True)   # gains group: logical (rows[, cols], T)
        self.x0 = x0.t().contiguous()                      # (n, N)
        self.weighting = torch.as_tensor(self.weighting_host, **f64)
        self.xref, self.uref = z(T + 1, n, N), z(T, m, N)
        self.xref[0] = self.x0                             # ilqr.py:59-60
        self.K = zg(m, n, T)
        if kff_init is None:
            kff_init = 1e-4 * np.random.randn(N, m, T)     # ilqr.py:70 (global NumPy RNG)
        self.kff = to_device(np.asarray(kff_init, dtype=np.float64).reshape(N, m, T), GL, dev)
        self.last_return = torch.full((N,), -float("inf"), **f64)   # ilqr.py:87
        self.lmbda = torch.full((N,), float(lmbda), **f64)
        self.dlmbda = torch.full((N,), float(dlmbda), **f64)
        self.alpha = torch.full((N,), float("nan"), **f64)
        self.status = torch.zeros(N, **i32)
        self.iters = torch.zeros(N, **i32)
        self.trace_len = torch.zeros(N, **i32)
        self.trace_cap = int(trace_cap) if trace_cap is not None else 101
        self.trace = z(self.trace_cap, N)
        self.keep_value = bool(keep_value)
        na = self.alphas.size
        # scratch
        self._A, self._B = zg(n, n, T), zg(n, m, T)
        self._Cxx, self._cx = zg(n, n, T + 1), zg(n, T + 1)
        self._Cuu, self._cu, self._Cxu = zg(m, m, T + 1), zg(m, T + 1), zg(n, m, T + 1)
        # candidate gains, three banks (include/trajopt_amd.h, trajopt_backward_io_t::out_bank): an accept records the
        # bank instead of copying - PROBLEM_MINOR (T, 3, n, m, N), PROBLEM_MAJOR (3, N, T, n, m)
        if GL == LAYOUT_PROBLEM_MINOR:
            self._cand_K, self._cand_kff = z(T, 3, n, m, N), z(T, 3, m, N)
        else:
            self._cand_K, self._cand_kff = z(3, N, T, n, m), z(3, N, T, m)
        # dV = [kff'qu, 1/2 kff'Quu kff] per problem belongs to the gains group: (2, N) for PROBLEM_MINOR, (N, 2) for MAJOR
        self._dV, self._g_norm = (z(2, N) if GL == LAYOUT_PROBLEM_MINOR else z(N, 2)), z(N)
        self._diverge, self._active, self._sel = torch.zeros(N, **i32), torch.zeros(N, **i32), torch.zeros(N, **i32)
        # candidates of every alpha: two banks of `na` planes + one plane nobody reads; the reference of problem p is
        # plane _ref_plane[p] (< 0: xref / uref), its accepted gains are bank _gains_bank[p] (< 0: K / kff).
        # Planes are time-blocked, (block of 8 steps, row, N, step in block): csrc/common.h
        self._cand_state, self._cand_action = z(2 * na + 1, (T + 8) // 8, n, N, 8), z(2 * na + 1, (T + 7) // 8, m, N, 8)
        self._ref_plane, self._gains_bank = torch.full((N,), -1, **i32), torch.full((N,), -1, **i32)
        self._out_bank = torch.zeros(N, **i32)
        self._cand_ret, self._cand_finite = z(na, N), torch.zeros((na, N), **i32)
        self._pending_list, self._pending_count = torch.zeros(N, **i32), torch.zeros(1, **i32)
        if self.keep_value:
            self.V, self.v = zg(n, n, T + 1), zg(n, T + 1)
            self.Qxx, self.Qux, self.Quu = zg(n, n, T), zg(m, n, T), zg(m, m, T)
            self.qx, self.qu = zg(n, T), zg(m, T)
            self._cand_V, self._cand_v = zg(n, n, T + 1), zg(n, T + 1)
            self._cand_Qxx, self._cand_Qux, self._cand_Quu = zg(n, n, T), zg(m, n, T), zg(m, m, T)
            self._cand_qx, self._cand_qu = zg(n, T), zg(m, T)

        o = ILQROptions()
        o.sched.min_lmbda, o.sched.max_lmbda, o.sched.mult_lmbda = float(min_lmbda), float(max_lmbda), float(mult_lmbda)
        o.tolfun, o.tolgrad, o.min_imp, o.reg = float(tolfun), float(tolgrad), float(min_imp), int(reg)
        o.nb_alphas = na
        for i, a in enumerate(self.alphas):
            o.alphas[i] = float(a)
        o.argmin_alpha, o.keep_value = int(bool(argmin_alpha)), int(self.keep_value)
        o.staged_line_search = int(staged_line_search)  # -1 auto, 0 one launch, 1 staged, 2 two passes (same results)
        o.no_fusion = int(bool(no_fusion))              # True: quadratise and backward pass always as two launches
        self._opt = o
        self._ws = self._make_ws()
        self._initialised = False
        self._trace_bound = 0
        self._lmbda0, self._dlmbda0 = float(lmbda), float(dlmbda)

    def reset(self, init_state, kff_init=None, generator=None):
        """Re-arm the solver for new initial states without reallocating anything: the state a FRESH
        iLQR(env, nb_steps, init_state) would have (ilqr.py:59-70,87) - the reference's MPC loops build a new
        solver at every control step (examples/ilqr/mpc/pendulum.py:22-27).  init_state: (N, dm_state) array or
        device tensor; kff_init as in the constructor.  generator: a torch.Generator of this device - with kff_init
        None the initial feed-forward 1e-4 * randn (ilqr.py:70) is then drawn ON THE DEVICE, straight into the gains
        layout (nothing crosses PCIe: with a device init_state the whole reset is device work); without it the draw comes
        from np.random as in the reference, 8 N m T bytes generated on the host and uploaded."""
        import torch
        N, n, m, T = self.nb_problems, self.dm_state, self.dm_act, self.nb_steps
        if torch.is_tensor(init_state):
            x0 = init_state.detach().to(device=self.device, dtype=torch.float64).reshape(N, n)
        else:
            x0 = torch.as_tensor(np.asarray(init_state, dtype=np.float64).reshape(N, n), device=self.device)
        self.x0.copy_(x0.t())
        self.xref.zero_()
        self.xref[0] = self.x0
        self.uref.zero_()
        self.K.zero_()
        if kff_init is None and generator is not None:
            self.kff.normal_(generator=generator).mul_(1e-4)   # iid: the layout does not matter
            kff_init = self.kff
        elif kff_init is None:
            kff_init = 1e-4 * np.random.randn(N, m, T)
        if kff_init is self.kff:
            pass
        elif torch.is_tensor(kff_init):   # logical (N, m, T), as the NumPy branch: permuted into the gains layout
            k = kff_init.detach().to(device=self.device, dtype=torch.float64).reshape(N, m, T)
            self.kff.copy_(k.permute(2, 1, 0) if self.gains_layout == LAYOUT_PROBLEM_MINOR else k.permute(0, 2, 1))
        else:
            self.kff.copy_(to_device(np.asarray(kff_init, dtype=np.float64).reshape(N, m, T), self.gains_layout,
                                     self.device))
        self.last_return.fill_(-float("inf"))
        self.lmbda.fill_(self._lmbda0)
        self.dlmbda.fill_(self._dlmbda0)
        self.alpha.fill_(float("nan"))
        self.status.zero_()
        self.iters.zero_()
        self.trace_len.zero_()
        self._trace_bound = 0
        self._initialised = False
        self._ref_plane.fill_(-1)   # reference and controller are the dense xref / uref / K / kff again
        self._gains_bank.fill_(-1)

    def _reserve_trace(self, nb_iter):
        """The device appends one return per accepted iteration (ilqr.py:219) and drops what does not fit: make room
        for `nb_iter` more entries first (upper bound kept on the host, so no synchronisation)."""
        import torch
        need = self._trace_bound + int(nb_iter)
        if need > self.trace_cap:
            cap = max(need, 2 * self.trace_cap)
            trace = torch.zeros((cap, self.nb_problems), dtype=torch.float64, device=self.device)
            trace[:self.trace_cap] = self.trace
            self.trace, self.trace_cap = trace, cap
            self._ws = self._make_ws()
        self._trace_bound = need

    def _make_ws(self):
        w = ILQRWorkspace()
        w.nb_steps, w.nb_problems, w.trace_cap = self.nb_steps, self.nb_problems, self.trace_cap
        w.gains_layout = self.gains_layout
        pub = ("x0", "weighting", "xref", "uref", "K", "kff", "last_return", "lmbda", "dlmbda", "alpha", "status",
               "iters", "trace_len", "trace")
        for k in pub:
            setattr(w, k, dptr(getattr(self, k)))
        for k in ("A", "B", "Cxx", "cx", "Cuu", "cu", "Cxu", "cand_K", "cand_kff", "dV", "g_norm", "diverge", "active",
                  "sel", "cand_state", "cand_action", "cand_ret", "cand_finite", "pending_list", "pending_count", "ref_plane",
                  "gains_bank", "out_bank"):
            setattr(w, k, dptr(getattr(self, "_" + k)))
        if self.keep_value:
            for k in ("V", "v", "Qxx", "Qux", "Quu", "qx", "qu"):
                setattr(w, k, dptr(getattr(self, k)))
                setattr(w, "cand_" + k, dptr(getattr(self, "_cand_" + k)))
        return w

    # ------------------------------------------------------------------------------------------
    def init_rollout(self):
        """ilqr.py:139-149."""
        check(lib().trajopt_ilqr_init_rollout(C.byref(self._struct), C.byref(self._opt), C.byref(self._ws), stream_ptr()))
        self._trace_bound = 1            # trace[0] = the initial return (ilqr.py:149)
        self._initialised = True

    def plan(self):
        """What an iteration launches: dict(structured_cost, fused_quad_bwd, staged_line_search, backward_kernel)."""
        pl = _lib.ILQRPlan()
        check(lib().trajopt_ilqr_plan(C.byref(self._struct), C.byref(self._opt), C.byref(self._ws), C.byref(pl)))
        return {k: int(getattr(pl, k)) for k, _ in pl._fields_}

    def iterate(self, nb_iter=1):
        """`nb_iter` iterations of ilqr.py:151-231 for every problem still running; asynchronous."""
        self._reserve_trace(nb_iter)
        try:
            check(lib().trajopt_ilqr_iterate(C.byref(self._struct), C.byref(self._opt), C.byref(self._ws), int(nb_iter),
                                             stream_ptr()))
        except _lib.TrajoptError:
            self._initialised = False   # a launch failed part-way: the workspace needs reset() before it is used again
            raise

    def run(self, nb_iter=25, verbose=False):
        """iLQR.run for the whole batch.  Returns the list of per-problem cost traces (ilqr.py:137,237)."""
        import torch
        if not self._initialised:
            self.init_rollout()
        self.iterate(nb_iter)
        torch.cuda.synchronize(self.device)
        tl = self.trace_len.cpu().numpy()
        tr = self.trace.cpu().numpy()
        if verbose:
            st = self.status.cpu().numpy()
            print("batched iLQR: %d problems, mean return %.6g, status counts %s" % (
                self.nb_problems, float(self.last_return.mean()), np.bincount(st, minlength=4).tolist()))
        return [tr[:tl[p], p].tolist() for p in range(self.nb_problems)]

    def forward_pass(self, K, kff, alphas=None, with_cost=True):
        """ilqr.py:89-102 for every (problem, alpha): -> state (na, T+1, n, N), action (na, T, m, N),
        cost (na, T+1, N), ret (na, N), finite (na, N) device tensors."""
        import torch
        al = self.alphas if alphas is None else np.atleast_1d(np.asarray(alphas, dtype=np.float64))
        na, n, m, T, N = al.size, self.dm_state, self.dm_act, self.nb_steps, self.nb_problems
        f64 = dict(dtype=torch.float64, device=self.device)
        state, action = torch.empty((na, T + 1, n, N), **f64), torch.empty((na, T, m, N), **f64)
        cost = torch.empty((na, T + 1, N), **f64) if with_cost else None
        ret = torch.empty((na, N), **f64)
        finite = torch.empty((na, N), dtype=torch.int32, device=self.device)
        arr = (C.c_double * na)(*[float(a) for a in al])
        if self.gains_layout != LAYOUT_PROBLEM_MINOR:  # the public rollout entry takes one layout for everything
            K = to_device(from_device(K, self.gains_layout), LAYOUT_PROBLEM_MINOR, self.device)
            kff = to_device(from_device(kff, self.gains_layout), LAYOUT_PROBLEM_MINOR, self.device)
        check(lib().trajopt_ilqr_forward_pass_batched(
            C.byref(self._struct), T, C.c_int64(N), LAYOUT_PROBLEM_MINOR, dptr(self.x0), dptr(K), dptr(kff),
            dptr(self.xref), dptr(self.uref), dptr(self.weighting), arr, na, None, dptr(state), dptr(action),
            dptr(cost), dptr(ret), dptr(finite), stream_ptr()))
        return state, action, cost, ret, finite

    def host(self, name):
        """Logical NumPy view (N, rows[, cols], T) of a per-step tensor, or the plain (N,) vector."""
        t = getattr(self, name)
        if t.dim() == 1:
            return t.cpu().numpy()
        return from_device(t, self.gains_layout if name in self.GAINS_GROUP else LAYOUT_PROBLEM_MINOR)


class iLQR:
    """Drop-in for trajopt.ilqr.iLQR (ilqr.py:10-237)."""

    def __init__(self, env, nb_steps, init_state, activation=None, slew_rate=False, action_penalty=False,
                 alphas=DEFAULT_ALPHAS, lmbda=1., dlmbda=1., min_lmbda=1e-6, max_lmbda=1e6, mult_lmbda=1.6,
                 tolfun=1e-6, tolgrad=1e-4, min_imp=0., reg=1):
        self.env = env
        self.env_dyn = self.env.unwrapped.dynamics
        self.env_cost = self.env.unwrapped.cost
        self.env_init = init_state
        self.ulim = self.env.action_space.high
        self.dm_state = self.env.observation_space.shape[0]
        self.dm_act = self.env.action_space.shape[0]
        self.nb_steps = nb_steps

        self.env.unwrapped.slew_rate = slew_rate
        if action_penalty is not None:
            self.env.unwrapped.uw = action_penalty * np.ones((self.dm_act,))

        self.alphas = alphas
        self.alpha = None
        self.lmbda, self.dlmbda = lmbda, dlmbda
        self.min_lmbda, self.max_lmbda, self.mult_lmbda = min_lmbda, max_lmbda, mult_lmbda
        self.reg = reg
        self.min_imp = min_imp
        self.tolfun, self.tolgrad = tolfun, tolgrad

        self.xref = np.zeros((self.dm_state, self.nb_steps + 1))
        self.xref[..., 0] = self.env_init
        self.uref = np.zeros((self.dm_act, self.nb_steps))

        self.vfunc = QuadraticStateValue(self.dm_state, self.nb_steps + 1)
        self.qfunc = QuadraticStateActionValue(self.dm_state, self.dm_act, self.nb_steps)
        self.dyn = AnalyticalLinearDynamics(self.env_dyn, self.dm_state, self.dm_act, self.nb_steps)
        self.ctl = LinearControl(self.dm_state, self.dm_act, self.nb_steps)
        self.ctl.kff = 1e-4 * np.random.randn(self.dm_act, self.nb_steps)

        self._activation = activation
        self.weighting = make_weighting(self.nb_steps, activation)
        self.cost = AnalyticalQuadraticCost(self.env_cost, self.dm_state, self.dm_act, self.nb_steps + 1)
        self.last_return = -np.inf
        self.status = STATUS_RUNNING
        self.failed = False

    # -- one-problem views of the batched machinery ---------------------------------------------
    def _batched(self):
        b = BatchedILQR(self.env, self.nb_steps, np.asarray(self.env_init, dtype=np.float64).reshape(1, -1),
                        kff_init=np.asarray(self.ctl.kff, dtype=np.float64)[None], alphas=self.alphas,
                        lmbda=self.lmbda, dlmbda=self.dlmbda, min_lmbda=self.min_lmbda, max_lmbda=self.max_lmbda,
                        mult_lmbda=self.mult_lmbda, tolfun=self.tolfun, tolgrad=self.tolgrad, min_imp=self.min_imp,
                        reg=self.reg, keep_value=True, apply_env_kwargs=False)
        b.weighting.copy_(b.weighting.new_tensor(self.weighting))
        import torch
        L = LAYOUT_PROBLEM_MINOR
        b.K.copy_(to_device(np.asarray(self.ctl.K, dtype=np.float64)[None], b.gains_layout, b.device))
        b.xref.copy_(to_device(np.asarray(self.xref, dtype=np.float64)[None], L, b.device))
        b.uref.copy_(to_device(np.asarray(self.uref, dtype=np.float64)[None], L, b.device))
        b.last_return.fill_(float(self.last_return))
        return b

    def forward_pass(self, ctl, alpha):
        """-> state (dm_state, T+1), action (dm_act, T), cost (T+1,)   (ilqr.py:89-102)"""
        if not self._has_device_model():
            from .hostenv import HostEnvSolver
            return HostEnvSolver(self).rollout(ctl, alpha)
        b = self._batched()
        L = LAYOUT_PROBLEM_MINOR
        K = to_device(np.asarray(ctl.K, dtype=np.float64)[None], b.gains_layout, b.device)
        kff = to_device(np.asarray(ctl.kff, dtype=np.float64)[None], b.gains_layout, b.device)
        state, action, cost, _, _ = b.forward_pass(K, kff, alphas=[alpha])
        return (np.ascontiguousarray(from_device(state[0], L)[0]), np.ascontiguousarray(from_device(action[0], L)[0]),
                cost[0, :, 0].cpu().numpy())

    def backward_pass(self):
        """-> (lc, xvalue, xuvalue, dV, diverge)   (ilqr.py:104-117)"""
        lc = LinearControl(self.dm_state, self.dm_act, self.nb_steps)
        xvalue = QuadraticStateValue(self.dm_state, self.nb_steps + 1)
        xuvalue = QuadraticStateActionValue(self.dm_state, self.dm_act, self.nb_steps)
        out, self.failed = core.backward_pass_flag(self.cost.Cxx, self.cost.cx, self.cost.Cuu, self.cost.cu,
                                                   self.cost.Cxu, self.dyn.A, self.dyn.B, self.lmbda, self.reg,
                                                   self.dm_state, self.dm_act, self.nb_steps)
        (xuvalue.Qxx, xuvalue.Qux, xuvalue.Quu, xuvalue.qx, xuvalue.qu, xvalue.V, xvalue.v, dV, lc.K, lc.kff,
         diverge) = out
        return lc, xvalue, xuvalue, dV, diverge

    def _has_device_model(self):
        return hasattr(self.env.unwrapped if hasattr(self.env, "unwrapped") else self.env, "struct")

    def run(self, nb_iter=25, verbose=False):
        """ilqr.py:136-237, executed by the batched device driver with N = 1 - or, for an env without a device model,
        by the host loop of hostenv.py around the GPU backward pass."""
        if not self._has_device_model():
            from .hostenv import HostEnvSolver
            _lib.require_gpu()
            trace = HostEnvSolver(self).run(nb_iter)
            if verbose:
                for i, r in enumerate(trace[1:]):
                    print("iter: ", i, " return: ", r)
            return trace
        b = self._batched()
        trace = b.run(nb_iter=nb_iter)[0]
        h = lambda k: np.ascontiguousarray(b.host(k)[0])
        self.xref, self.uref = h("xref"), h("uref")
        self.last_return = float(b.last_return[0])
        self.lmbda, self.dlmbda = float(b.lmbda[0]), float(b.dlmbda[0])
        a = float(b.alpha[0])
        self.alpha = None if a != a else a
        self.status = int(b.status[0])
        if len(trace) > 1:  # at least one accepted iteration: ilqr.py:214-217
            self.ctl = LinearControl(self.dm_state, self.dm_act, self.nb_steps)
            self.ctl.K, self.ctl.kff = h("K"), h("kff")
            self.vfunc = QuadraticStateValue(self.dm_state, self.nb_steps + 1)
            self.vfunc.V, self.vfunc.v = h("V"), h("v")
            self.qfunc = QuadraticStateActionValue(self.dm_state, self.dm_act, self.nb_steps)
            self.qfunc.Qxx, self.qfunc.Qux, self.qfunc.Quu = h("Qxx"), h("Qux"), h("Quu")
            self.qfunc.qx, self.qfunc.qu = h("qx"), h("qu")
        if verbose:
            for i, r in enumerate(trace[1:]):
                print("iter: ", i, " return: ", r)
        return trace
