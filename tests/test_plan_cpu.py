"""Host logic of trajopt_ilqr_plan (include/trajopt_amd.h): which kernels and which line-search schedule an iteration
of ilqr.py:151-231 takes for a shape and batch size.  No GPU: the workspace pointers are never dereferenced."""
import ctypes as C

import numpy as np
import pytest

from trajopt_amd import _lib, envs


def _plan(name, N, T=500, nb_alphas=11, gains_layout=1, **opt_kw):
    L = _lib.lib()
    env = envs.make(name, **({"dt": 0.02} if name != "pendulum" else {"dt": 0.05}))
    for k in ("slew_rate", "periodic"):
        if k in opt_kw:
            setattr(env, k, opt_kw.pop(k))
    st = env.struct()
    ws = _lib.ILQRWorkspace()
    for k, t in ws._fields_:
        if t is _lib._dp:
            setattr(ws, k, C.cast(C.c_void_p(8), _lib._dp))     # non-null, never read by the plan
    ws.nb_steps, ws.nb_problems, ws.gains_layout, ws.trace_cap = T, N, gains_layout, 4
    o = _lib.ILQROptions()
    o.nb_alphas, o.reg = nb_alphas, 1
    for i in range(nb_alphas):
        o.alphas[i] = 10.0 ** (-0.3 * i)
    o.staged_line_search = opt_kw.pop("staged_line_search", -1)
    o.argmin_alpha = opt_kw.pop("argmin_alpha", 0)
    assert not opt_kw
    pl = _lib.ILQRPlan()
    _lib.check(L.trajopt_ilqr_plan(C.byref(st), C.byref(o), C.byref(ws), C.byref(pl)))
    return {k: int(getattr(pl, k)) for k, _ in pl._fields_}


def test_line_search_schedule_by_batch_size():
    # small batches: every (problem, alpha) in one launch, the lone-wave rollout form
    p = _plan("cartpole", 4096)
    assert p["staged_line_search"] == 0 and p["rollout_straight_line"] == 1 and p["fused_quad_bwd"] == 1
    # from 1.4 waves per SIMD on (dx + du <= 5): returns first, the accepted alpha again
    assert _plan("cartpole", 8191)["staged_line_search"] == 0
    assert _plan("cartpole", 8192)["staged_line_search"] == 2
    # quadratisation + backward pass in one launch: four problems per wave below 32 768 problems, lane = problem from there
    assert _plan("cartpole", 16384)["fused_quad_bwd"] == 1 and _plan("cartpole", 32767)["fused_quad_bwd"] == 1
    assert _plan("cartpole", 32768)["fused_quad_bwd"] == 2
    assert _plan("double_cartpole", 8192)["fused_quad_bwd"] == 0
    p = _plan("cartpole", 65536)
    assert p["staged_line_search"] == 2 and p["rollout_straight_line"] == 0 and p["fused_quad_bwd"] == 2
    assert _plan("pendulum", 65536, T=100)["staged_line_search"] == 2
    # larger models: from 6 waves per SIMD on
    assert _plan("double_cartpole", 8192)["staged_line_search"] == 0
    assert _plan("double_cartpole", 32768)["staged_line_search"] == 0
    assert _plan("double_cartpole", 36000)["staged_line_search"] == 2
    assert _plan("quadrotor", 4096, gains_layout=0)["staged_line_search"] == 0


def test_line_search_schedule_on_request_and_fallbacks():
    assert _plan("cartpole", 256, staged_line_search=2)["staged_line_search"] == 2
    assert _plan("cartpole", 256, staged_line_search=1)["staged_line_search"] == 1
    assert _plan("cartpole", 65536, staged_line_search=0)["staged_line_search"] == 0
    # a single alpha: nothing to schedule
    assert _plan("cartpole", 65536, nb_alphas=1)["staged_line_search"] == 0
    assert _plan("cartpole", 256, nb_alphas=1, staged_line_search=2)["staged_line_search"] == 0
    # argmin needs every alpha: never staged, two passes are fine
    assert _plan("cartpole", 256, staged_line_search=1, argmin_alpha=1)["staged_line_search"] == 0
    assert _plan("cartpole", 65536, argmin_alpha=1)["staged_line_search"] == 2
    # no returns-only rollout for the slew-rate cost / wrapped angles: staged from 16 waves per SIMD on, else one launch
    assert _plan("pendulum", 65536, T=100, slew_rate=True)["staged_line_search"] == 0
    assert _plan("pendulum", 65536, T=100, periodic=True, staged_line_search=2)["staged_line_search"] == 0
    assert _plan("pendulum", 131072, T=100, slew_rate=True)["staged_line_search"] == 1


def test_plan_rejects_an_incomplete_workspace():
    L = _lib.lib()
    st = envs.make("cartpole", dt=0.02).struct()
    ws, o, pl = _lib.ILQRWorkspace(), _lib.ILQROptions(), _lib.ILQRPlan()
    ws.nb_steps, ws.nb_problems, ws.gains_layout, ws.trace_cap = 10, 4, 1, 4
    o.nb_alphas = 2
    assert L.trajopt_ilqr_plan(C.byref(st), C.byref(o), C.byref(ws), C.byref(pl)) == -1   # TRAJOPT_ERR_BAD_ARG


def test_sub_batch_rule_and_split():
    """trajopt_amd/ilqr/streamed.py: how many sub-batches a batch is cut into by default (the measured table in its
    docstring) and how the problems are dealt out - contiguous, as equal as possible, every problem exactly once."""
    from trajopt_amd.ilqr.streamed import auto_sub_batches, split_sizes
    assert auto_sub_batches(4, 1, 4096) == 1 and auto_sub_batches(4, 1, 16384) == 1 and auto_sub_batches(4, 1, 65536) == 1
    assert auto_sub_batches(2, 1, 65536) == 1 and auto_sub_batches(2, 1, 1) == 1
    assert auto_sub_batches(6, 1, 2048) == 1 and auto_sub_batches(6, 1, 4096) == 2 and auto_sub_batches(6, 1, 8192) == 2 and auto_sub_batches(6, 1, 16384) == 4
    assert auto_sub_batches(6, 1, 65536) == 1
    assert auto_sub_batches(12, 4, 4096) == 4 and auto_sub_batches(12, 4, 512) == 1
    for N, S in ((301, 3), (8192, 2), (7, 7), (10, 4), (1, 1)):
        sizes = split_sizes(N, S)
        assert len(sizes) == S and sum(sizes) == N and max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
        assert sizes == sorted(sizes, reverse=True)
