"""GPU parity of env models, rollout, quadratisation and the batched iLQR driver (rows A3-A8, A10)
against the CPU oracle, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from tests.util import TOL, assert_close, rel_err

pytestmark = pytest.mark.gpu

ENVS = ["pendulum", "cartpole", "double_cartpole", "quadrotor", "lqr", "double_pendulum", "robot", "lqr_v1", "lqr_v2", "quad_pendulum",
        # the ...WithCartesianCost classes: nonlinear cost features, frozen-Jacobian derivatives
        "pendulum_cartesian", "cartpole_cartesian", "double_cartpole_cartesian", "double_pendulum_cartesian",
        "quad_pendulum_cartesian"]


def _base(name):
    return name[:-len("_cartesian")] if name.endswith("_cartesian") else name


def _pair(name, **kw):
    """(product env, oracle env struct) with identical parameters."""
    from oracle import oracle as O
    from trajopt_amd import envs
    pe = envs.make(name, **kw)
    okw = dict(kw)
    if name != _base(name):
        name, okw["cartesian"] = _base(name), True
    if "uw" in okw:
        okw["uw"] = list(np.asarray(okw["uw"], dtype=float) * np.ones(pe.dm_act))
    oe = O.make_env(name, **okw)
    return pe, oe


def _rand_state(name, rng, N):
    n = {"pendulum": 2, "cartpole": 4, "double_cartpole": 6, "quadrotor": 12, "lqr": 2, "double_pendulum": 4, "robot": 4,
         "lqr_v1": 2, "lqr_v2": 2, "quad_pendulum": 8}[_base(name)]
    x = rng.standard_normal((N, n))
    if name == "quadrotor":
        x[:, 3:6] *= 0.3  # keep away from the Euler singularity
    return x


def test_env_struct_matches_oracle_struct():
    for name in ENVS:
        for kw in ({}, dict(dt=0.05, uw=[1e-2] * {"quadrotor": 4, "double_pendulum": 2, "robot": 2, "quad_pendulum": 4}.get(_base(name), 1), slew_rate=True, periodic=True)):
            pe, oe = _pair(name, **kw)
            assert bytes(pe.struct()) == bytes(oe), name


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("name", ENVS)
def test_env_step_matches_oracle(oracle, name, layout):
    import torch
    from trajopt_amd import _lib, layout as L
    pe, oe = _pair(name, slew_rate=True, periodic=(_base(name) in ("pendulum", "cartpole", "double_cartpole", "double_pendulum", "quad_pendulum")))
    rng = np.random.default_rng(3)
    N = 130
    n, m = pe.dm_state, pe.dm_act
    x = 3.0 * _rand_state(name, rng, N)
    u = 8.0 * rng.standard_normal((N, m))          # some beyond umax: exercises the clip
    ul = rng.standard_normal((N, m))
    a = rng.uniform(0.1, 2.0, N)
    a[::7] = 0.0                                   # `if a:` drops the state cost
    e = pe.struct()
    dx, du, dul = (torch.as_tensor(np.ascontiguousarray(v if layout == 0 else v.T), device="cuda")
                   for v in (x, u, ul))
    da = torch.as_tensor(a, device="cuda")
    xn = torch.empty_like(dx)
    cost = torch.empty(N, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().trajopt_env_step_batched(C.byref(e), _lib.dptr(dx), _lib.dptr(du), _lib.dptr(dul),
                                                   _lib.dptr(da), C.c_int64(N), layout, _lib.dptr(xn), _lib.dptr(cost),
                                                   _lib.stream_ptr()))
    torch.cuda.synchronize()
    xn = xn.cpu().numpy() if layout == 0 else xn.cpu().numpy().T
    cost = cost.cpu().numpy()
    for p in range(N):
        assert_close(xn[p], oracle.env_dynamics(oe, x[p], u[p]), "%s xn[%d]" % (name, p), tol=1e-12)
        c = oracle.env_cost(oe, x[p], u[p], ul[p], a[p])
        assert abs(cost[p] - c) <= 1e-12 * max(abs(c), 1e-300), (name, p, cost[p], c)


@pytest.mark.parametrize("name", ["pendulum", "cartpole", "double_cartpole", "double_pendulum"])
def test_env_step_at_large_angles(oracle, name):
    """The device sin / cos has no big-argument path (env_models.h: reduction by pi with one FMA per Cody-Waite term, valid
    to |x| ~ 2^51): angles of 1e2 .. 1e12 rad must give the oracle's (libm's) step.  The comparison itself is conditioned
    by the angle's own ulp - a fused and an unfused `x + h k` differ by one ulp of x, and sin moves by as much - so the bar
    is 1e-12 + 64 ulp(|angle|); a reduction that loses the argument (what dropping a Payne-Hanek path naively would do)
    is off by O(1)."""
    import torch
    from trajopt_amd import _lib
    pe, oe = _pair(name)
    rng = np.random.default_rng(11)
    N = 260
    n, m = pe.dm_state, pe.dm_act
    x = _rand_state(name, rng, N)
    ang = {"pendulum": [0], "cartpole": [1], "double_cartpole": [1, 2], "double_pendulum": [0, 1]}[name]
    mag = 10.0 ** rng.uniform(2.0, 12.0, (N, len(ang))) * rng.choice([-1.0, 1.0], (N, len(ang)))
    x[:, ang] = mag
    u = rng.standard_normal((N, m))
    e = pe.struct()
    dx, du = (torch.as_tensor(np.ascontiguousarray(v), device="cuda") for v in (x, u))
    da = torch.ones(N, dtype=torch.float64, device="cuda")
    xn = torch.empty_like(dx)
    cost = torch.empty(N, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().trajopt_env_step_batched(C.byref(e), _lib.dptr(dx), _lib.dptr(du), _lib.dptr(du), _lib.dptr(da),
                                                   C.c_int64(N), 0, _lib.dptr(xn), _lib.dptr(cost), _lib.stream_ptr()))
    torch.cuda.synchronize()
    xn = xn.cpu().numpy()
    assert np.all(np.isfinite(xn))
    worst = 0.0
    for p in range(N):
        ref = oracle.env_dynamics(oe, x[p], u[p])
        tol = 1e-12 + 64.0 * np.spacing(np.max(np.abs(mag[p])))
        err = np.max(np.abs(xn[p] - ref) / np.maximum(np.abs(ref), 1.0))
        worst = max(worst, err / tol)
        assert err <= tol, (name, p, mag[p], err, tol)


def _rollout_inputs(name, rng, N, T, n, m):
    xref = _rand_state(name, rng, N)[:, :, None] + 0.1 * rng.standard_normal((N, n, T + 1))
    uref = rng.standard_normal((N, m, T))
    # the quadrotor's attitude loop (inertia 4e-3) amplifies round-off e^(O(100 t)) under random gains:
    # keep its rollouts tame so that 1e-9 tests the arithmetic, not the Lyapunov exponent
    gs = 0.01 if name == "quadrotor" else 0.3
    K = gs * rng.standard_normal((N, m, n, T))
    kff = (0.1 if name == "quadrotor" else 1.0) * rng.standard_normal((N, m, T))
    if name == "quadrotor":
        uref = 1.2 + 0.05 * uref
    if name == "robot":   # steering angle: tan(phi) / l = 10 tan(phi) turns round-off into O(1) near its poles
        K[:, 1], kff[:, 1], uref[:, 1] = 0.03 * K[:, 1], 0.1 * kff[:, 1], 0.2 * uref[:, 1]
    x0 = xref[:, :, 0] + 0.05 * rng.standard_normal((N, n))
    return x0, K, kff, xref, uref


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("name", ENVS)
def test_forward_pass_matches_oracle(oracle, name, layout):
    """ilqr.py:89-102 for several alphas at once; trajectory cost within 1e-9 (north_star)."""
    import torch
    from trajopt_amd import _lib, layout as L
    pe, oe = _pair(name, dt=0.02 if name != "lqr" else None)
    rng = np.random.default_rng(11)
    N, T = 70, 40
    n, m = pe.dm_state, pe.dm_act
    x0, K, kff, xref, uref = _rollout_inputs(name, rng, N, T, n, m)
    w = rng.uniform(0.5, 1.5, T + 1)
    alphas = np.array([1.0, 0.5, 0.1])
    e = pe.struct()
    d = {k: L.to_device(v, layout) for k, v in dict(K=K, kff=kff, xref=xref, uref=uref).items()}
    dx0 = torch.as_tensor(np.ascontiguousarray(x0 if layout == 0 else x0.T), device="cuda")
    dw = torch.as_tensor(w, device="cuda")
    na = len(alphas)
    f64 = dict(dtype=torch.float64, device="cuda")
    state = torch.empty((na,) + L.device_shape((N, n, T + 1), layout), **f64)
    action = torch.empty((na,) + L.device_shape((N, m, T), layout), **f64)
    cost = torch.empty((na,) + L.device_shape((N, T + 1), layout), **f64)
    ret = torch.empty((na, N), **f64)
    fin = torch.empty((na, N), dtype=torch.int32, device="cuda")
    arr = (C.c_double * na)(*alphas)
    _lib.check(_lib.lib().trajopt_ilqr_forward_pass_batched(
        C.byref(e), T, C.c_int64(N), layout, _lib.dptr(dx0), _lib.dptr(d["K"]), _lib.dptr(d["kff"]),
        _lib.dptr(d["xref"]), _lib.dptr(d["uref"]), _lib.dptr(dw), arr, na, None, _lib.dptr(state), _lib.dptr(action),
        _lib.dptr(cost), _lib.dptr(ret), _lib.dptr(fin), _lib.stream_ptr()))
    torch.cuda.synchronize()
    for ai, alpha in enumerate(alphas):
        st, ac, co = (L.from_device(t[ai], layout) for t in (state, action, cost))
        for p in range(0, N, 3):
            rs, ra, rc = oracle.forward_pass(oe, T, x0[p], K[p], kff[p], alpha, xref[p], uref[p], w)
            assert_close(st[p], rs, "state")
            assert_close(ac[p], ra, "action")
            assert_close(co[p], rc, "cost")
            assert abs(ret[ai, p].item() - rc.sum()) <= TOL * abs(rc.sum())
            assert fin[ai, p].item() == int(np.all(rs < 1e8))


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("convention", [0, 1])
@pytest.mark.parametrize("name", ENVS)
def test_quadratize_matches_oracle(oracle, name, layout, convention):
    """objects.py:68-78,114-117 (and riccati/objects.py:54-65,102-108), incl. saturated inputs (clip mask),
    slew-rate cost, periodic wrap and zero weights."""
    import torch
    from trajopt_amd import _lib, layout as L
    periodic = _base(name) in ("pendulum", "cartpole", "double_cartpole")
    pe, oe = _pair(name, slew_rate=True, periodic=periodic)
    rng = np.random.default_rng(5)
    N, T = 66, 12
    n, m = pe.dm_state, pe.dm_act
    xref = 2.0 * _rand_state(name, rng, N)[:, :, None] + 0.3 * rng.standard_normal((N, n, T + 1))
    uref = 6.0 * rng.standard_normal((N, m, T))    # some saturated -> B columns vanish
    uref[:, :, 3] = np.asarray(pe.umax) if np.all(np.isfinite(pe.umax)) else 1.0   # exactly on the bound
    if name == "robot":
        uref[:, 1] = 0.4 * np.tanh(uref[:, 1])    # steering angle away from tan's poles
    w = rng.uniform(0.5, 1.5, T + 1)
    w[2] = 0.0
    e = pe.struct()
    dx, du = L.to_device(xref, layout), L.to_device(uref, layout)
    dw = torch.as_tensor(w, device="cuda")
    shp = dict(A=(N, n, n, T), B=(N, n, m, T), c=(N, n, T), Cxx=(N, n, n, T + 1), cx=(N, n, T + 1),
               Cuu=(N, m, m, T + 1), cu=(N, m, T + 1), Cxu=(N, n, m, T + 1))
    o = {k: L.empty(s, layout, zero=True) for k, s in shp.items()}
    _lib.check(_lib.lib().trajopt_quadratize_batched(
        C.byref(e), T, C.c_int64(N), layout, convention, _lib.dptr(dx), _lib.dptr(du), _lib.dptr(dw), None,
        _lib.dptr(o["A"]), _lib.dptr(o["B"]), _lib.dptr(o["c"]), _lib.dptr(o["Cxx"]), _lib.dptr(o["cx"]),
        _lib.dptr(o["Cuu"]), _lib.dptr(o["cu"]), _lib.dptr(o["Cxu"]), _lib.stream_ptr()))
    torch.cuda.synchronize()
    h = {k: L.from_device(v, layout) for k, v in o.items()}
    keys = list(shp) if convention == 1 else [k for k in shp if k != "c"]
    for p in range(0, N, 5):
        ref = oracle.quadratize(oe, T, xref[p], uref[p], w, riccati=bool(convention))
        for k in keys:
            den = max(np.abs(ref[k]).max(), 1.0)
            assert np.abs(h[k][p] - ref[k]).max() <= 1e-10 * den, "%s %s[%d]" % (name, k, p)


def _solve_pair(oracle, name, N, T, iters, seed=1337, scale=1e-2, gains_layout=None, **kw):
    from trajopt_amd.ilqr import BatchedILQR
    pe, oe = _pair(name, **kw)
    rng = np.random.Generator(np.random.PCG64(seed))
    x0 = np.asarray(pe.x0)[None] + scale * rng.standard_normal((N, pe.dm_state))
    kff0 = 1e-4 * rng.standard_normal((N, pe.dm_act, T))
    s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, keep_value=True, apply_env_kwargs=False,
                    gains_layout=gains_layout)
    traces = s.run(nb_iter=iters)
    refs = [oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=iters) for p in range(N)]
    return s, traces, refs


@pytest.mark.parametrize("name,T,iters,kw", [
    ("pendulum", 100, 5, dict(dt=0.05, uw=[1e-5])),           # config C1's shape (examples/ilqr/mpc/pendulum.py)
    ("cartpole", 60, 4, dict(dt=0.02, uw=[1e-5])),
    ("double_cartpole", 50, 3, dict(dt=0.01, uw=[1e-5])),
    ("quadrotor", 40, 3, dict(dt=0.02)),                      # gains group defaults to PROBLEM_MAJOR (dx > 6)
    ("lqr", 60, 5, dict()),
    ("double_pendulum", 50, 4, dict(dt=0.02)),                # a further env model, dx = 4, du = 2
    ("robot", 50, 4, dict()),                                 # 8f.3: dx = 4, du = 2, RK4 by -dt, cost x dt
    ("quad_pendulum", 30, 2, dict(dt=0.01)),                  # 8f.3: dx = 8, du = 4 (matrix-core backward kernel); chaotic: a 3rd iteration amplifies round-off into a different accept decision
    ("lqr_v1", 60, 4, dict()),
    ("lqr_v2", 60, 4, dict()),
    ("cartpole_cartesian", 60, 4, dict(dt=0.02, uw=[1e-5])),  # 8f.3: Cartesian cost - dense Cxx, no structured cost
    ("pendulum_cartesian", 80, 4, dict(dt=0.05, uw=[1e-5])),
    ("double_cartpole_cartesian", 40, 3, dict(dt=0.01, uw=[1e-5])),
    ("quad_pendulum_cartesian", 22, 2, dict(dt=0.01)),       # dx + du = 12 on the GENERAL rollout (cost features): half-block stage, partial last half
    ("cartpole", 40, 4, dict(dt=0.02, uw=[1e-5], gains_layout=0)),          # forced PROBLEM_MAJOR gains group
    ("double_cartpole", 30, 3, dict(dt=0.01, uw=[1e-5], gains_layout=0)),
    ("quadrotor", 30, 2, dict(dt=0.02, gains_layout=1)),                    # forced PROBLEM_MINOR
])
def test_batched_driver_matches_oracle_driver(oracle, name, T, iters, kw):
    """ilqr.py:136-237 per problem: same accepted iterates, lambda schedule, traces and stop reasons."""
    N = 37
    s, traces, refs = _solve_pair(oracle, name, N, T, iters, **kw)
    lam, dlam, st, it = s.host("lmbda"), s.host("dlmbda"), s.host("status"), s.host("iters")
    xref, uref, K, kff, V, v = (s.host(k) for k in ("xref", "uref", "K", "kff", "V", "v"))
    ret = s.host("last_return")
    for p in range(N):
        r = refs[p]
        assert len(traces[p]) == len(r["trace"]), (p, traces[p], r["trace"])
        assert_close(np.array(traces[p]), r["trace"], "trace[%d]" % p)
        assert abs(ret[p] - r["last_return"]) <= TOL * abs(r["last_return"])
        assert lam[p] == pytest.approx(r["lmbda"], rel=1e-14) and dlam[p] == pytest.approx(r["dlmbda"], rel=1e-14)
        assert st[p] == r["status"] and it[p] == r["iters"]
        assert_close(xref[p], r["xref"], "xref", tol=1e-8)
        assert_close(uref[p], r["uref"], "uref", tol=1e-8)
        if len(r["trace"]) > 1:
            assert_close(K[p], r["K"], "K", tol=1e-8)
            assert_close(kff[p], r["kff"], "kff", tol=1e-7)
            assert_close(V[p], r["V"], "V", tol=1e-8)
            assert_close(v[p], r["v"], "v", tol=1e-8)


def test_lq_problem_converges_in_one_step_with_exact_expected_reduction(oracle):
    """Known answer (SURVEY 8c iii): on an LQ problem with lmbda -> 0 and alpha = 1 the actual cost drop equals
    -(dV[0] + dV[1]) and the next iteration stops on the gradient-norm rule."""
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    env = envs.make("lqr")
    N, T = 8, 30
    rng = np.random.default_rng(0)
    x0 = np.asarray(env.x0)[None] + rng.standard_normal((N, 2))
    s = BatchedILQR(env, T, x0, kff_init=np.zeros((N, 1, T)), action_penalty=None, apply_env_kwargs=False,
                    lmbda=0.0, min_lmbda=0.0)
    s.init_rollout()
    r0 = s.host("last_return").copy()
    s.iterate(1)
    r1 = s.host("last_return")
    dV = s._dV.cpu().numpy()
    assert np.all(s.host("alpha") == 1.0)
    np.testing.assert_allclose(r0 - r1, -(dV[0] + dV[1]), rtol=1e-8)
    s.iterate(1)   # converged: the gradient-norm rule fires before any rollout (ilqr.py:176-180)
    assert np.all(s.host("status") == 1)


def test_argmin_alpha_option_never_worse_than_first_accept(oracle):
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    rng = np.random.default_rng(4)
    N, T = 64, 50
    x0 = np.array([0., np.pi, 0., 0.])[None] + 1e-2 * rng.standard_normal((N, 4))
    kff0 = 1e-4 * rng.standard_normal((N, 1, T))
    out = []
    for argmin in (False, True):
        env = envs.make("cartpole", dt=0.02)
        s = BatchedILQR(env, T, x0, kff_init=kff0, action_penalty=1e-5, argmin_alpha=argmin)
        s.run(nb_iter=1)
        out.append(s.host("last_return").copy())
    assert np.all(out[1] <= out[0] * (1 + 1e-12))


def test_ilqr_class_is_a_drop_in_for_the_reference_usage(oracle):
    """The call pattern of /root/reference/examples/ilqr/mpc/pendulum.py:18-31 (one MPC step)."""
    from trajopt_amd import envs
    from trajopt_amd.ilqr import iLQR
    env = envs.make('Pendulum-TO-v0')
    env.unwrapped.dt = 0.05
    np.random.seed(1337)
    state = np.array([np.pi, 0.])
    solver = iLQR(env, init_state=state, nb_steps=20, action_penalty=np.array([1e-5]))
    kff0 = solver.ctl.kff.copy()
    trace = solver.run(nb_iter=10, verbose=False)
    oe = oracle.make_env("pendulum", dt=0.05, uw=[1e-5])
    ref = oracle.ilqr_run(oe, 20, state, kff0, nb_iter=10)
    assert_close(np.array(trace), ref["trace"], "trace")
    assert solver.uref.shape == (1, 20) and solver.xref.shape == (2, 21)
    assert_close(solver.uref, ref["uref"], "uref", tol=1e-8)
    assert_close(solver.ctl.K, ref["K"], "K", tol=1e-8)
    assert_close(solver.vfunc.V, ref["V"], "V", tol=1e-8)
    assert solver.lmbda == pytest.approx(ref["lmbda"], rel=1e-14)
    # the individual methods keep the reference's signatures and shapes
    st, ac, co = solver.forward_pass(solver.ctl, 0.5)
    rs, ra, rc = oracle.forward_pass(oe, 20, state, solver.ctl.K, solver.ctl.kff, 0.5, solver.xref, solver.uref,
                                     np.ones(21))
    assert st.shape == (2, 21) and ac.shape == (1, 20) and co.shape == (21,)
    assert_close(st, rs, "state"); assert_close(co, rc, "cost")
    solver.dyn.taylor_expansion(solver.xref, solver.uref)
    solver.cost.taylor_expansion(solver.xref, solver.uref, solver.weighting)
    lc, xvalue, xuvalue, dV, diverge = solver.backward_pass()
    q = oracle.quadratize(oe, 20, solver.xref, solver.uref, np.ones(21))
    rb, _ = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], solver.lmbda, 1, 2, 1, 20)
    assert_close(lc.K, rb[8], "K"); assert_close(xvalue.V, rb[5], "V"); assert_close(xuvalue.Qux, rb[1], "Qux")
    assert_close(dV, rb[7], "dV"); assert diverge == 0


def test_riccati_class_on_the_lqr_env_reaches_the_dare_gain(oracle):
    """examples/riccati/lqr.py:9-18 call pattern; known answer: K_0 -> -(R + B'PB)^-1 B'PA (SURVEY 8c i)."""
    import scipy.linalg
    from trajopt_amd import envs
    from trajopt_amd.riccati import Riccati
    env = envs.make('LQR-TO-v0')
    env.seed(1337)
    T = 150
    solver = Riccati(env, nb_steps=T, init_state=env.init())
    ret = solver.run()
    assert np.isfinite(ret)
    A, B = env.A, env.B
    Q, R = env.dt * np.diag(env.gw), env.dt * np.diag(env.uw)
    P = scipy.linalg.solve_discrete_are(A, B, Q, R)
    Kinf = -np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)
    np.testing.assert_allclose(solver.ctl.K[..., 0], Kinf, rtol=1e-9)
    np.testing.assert_allclose(solver.vfunc.V[..., 0], P, rtol=1e-8)  # scipy DARE accuracy


def test_batched_mpc_matches_the_reference_loop(oracle):
    """examples/ilqr/mpc/pendulum.py:18-31 for N systems at once: a fresh horizon solve per control step, apply
    uref[:, 0], step the plant.  Checked against the same loop on the oracle (deterministic plant + injected noise)."""
    from trajopt_amd import envs
    from trajopt_amd.mpc import BatchedMPC
    N, horizon, steps, iters = 9, 20, 6, 5
    rng = np.random.Generator(np.random.PCG64(7))
    env = envs.make("Pendulum-TO-v0")
    env.unwrapped.dt = 0.05
    x0 = np.array([np.pi, 0.])[None] + 0.05 * rng.standard_normal((N, 2))
    kff = 1e-4 * rng.standard_normal((steps, N, 1, horizon))
    noise = 1e-3 * rng.standard_normal((N, 2, steps))
    mpc = BatchedMPC(env, horizon, x0, nb_iter=iters, action_penalty=np.array([1e-5]))
    out = mpc.run(steps, process_noise=noise, kff_inits=kff)
    oe = oracle.make_env("pendulum", dt=0.05, uw=[1e-5])
    for p in range(N):
        x = x0[p].copy()
        for t in range(steps):
            r = oracle.ilqr_run(oe, horizon, x, kff[t, p], nb_iter=iters)
            u = r["uref"][:, 0]
            assert_close(out["state"][p, :, t], x, "state[%d,%d]" % (p, t), tol=1e-7)
            assert abs(out["action"][p, 0, t] - u[0]) <= 1e-7 * max(1.0, abs(u[0])), (p, t)
            assert abs(out["cost"][p, t] - r["last_return"]) <= 1e-8 * abs(r["last_return"])
            x = oracle.env_dynamics(oe, x, u) + noise[p, :, t]


def test_batched_mpc_with_shift_warm_start_matches_the_same_loop_on_the_oracle(oracle):
    """SURVEY 8f.1's receding-horizon loop with the shift warm start (BatchedMPC.run(warm_start=True)): from the second
    control step on every solve starts from the previous plan shifted by one step (last action repeated) as its initial
    feed-forward - against the oracle's iLQR run in the same loop."""
    from trajopt_amd import envs
    from trajopt_amd.mpc import BatchedMPC
    N, horizon, steps, iters = 7, 20, 6, 4
    rng = np.random.Generator(np.random.PCG64(17))
    env = envs.make("Pendulum-TO-v0")
    env.unwrapped.dt = 0.05
    x0 = np.array([np.pi, 0.])[None] + 0.05 * rng.standard_normal((N, 2))
    kff = 1e-4 * rng.standard_normal((1, N, 1, horizon))
    noise = 1e-3 * rng.standard_normal((N, 2, steps))
    mpc = BatchedMPC(env, horizon, x0, nb_iter=iters, action_penalty=np.array([1e-5]))
    out = mpc.run(steps, process_noise=noise, kff_inits=np.repeat(kff, steps, axis=0), warm_start=True)
    oe = oracle.make_env("pendulum", dt=0.05, uw=[1e-5])
    for p in range(N):
        x = x0[p].copy()
        k0 = kff[0, p]
        for t in range(steps):
            r = oracle.ilqr_run(oe, horizon, x, k0, nb_iter=iters)
            u = r["uref"][:, 0]
            assert_close(out["state"][p, :, t], x, "state[%d,%d]" % (p, t), tol=1e-7)
            assert abs(out["action"][p, 0, t] - u[0]) <= 1e-7 * max(1.0, abs(u[0])), (p, t)
            assert abs(out["cost"][p, t] - r["last_return"]) <= 1e-8 * abs(r["last_return"])
            x = oracle.env_dynamics(oe, x, u) + noise[p, :, t]
            k0 = np.concatenate((r["uref"][:, 1:], r["uref"][:, -1:]), axis=1)


@pytest.mark.parametrize("rng_mode", ["given", "device"])
def test_batched_mpc_as_a_hip_graph_is_bitwise_the_eager_loop(rng_mode):
    """BatchedMPC.run(graph=True): one control step (re-arm, initial rollout, iterations, plant step) captured once as a
    HIP graph and replayed per step must give exactly what the eager loop of launches gives - with the kff noise and the
    process noise handed in, and with both drawn by device generators inside the captured step."""
    from trajopt_amd import envs
    from trajopt_amd.mpc import BatchedMPC
    N, horizon, steps, iters = 70, 24, 5, 4
    rng = np.random.Generator(np.random.PCG64(9))
    env = envs.make("cartpole", dt=0.02)
    x0 = np.asarray(env.x0)[None] + 0.1 * rng.standard_normal((N, 4))
    outs = []
    for graph in (False, True):
        mpc = BatchedMPC(env, horizon, x0, nb_iter=iters, action_penalty=np.array([1e-5]))
        if rng_mode == "given":
            kff = 1e-4 * np.random.Generator(np.random.PCG64(3)).standard_normal((steps, N, 1, horizon))
            noise = 1e-3 * np.random.Generator(np.random.PCG64(4)).standard_normal((N, 4, steps))
            outs.append(mpc.run(steps, process_noise=noise, kff_inits=kff, graph=graph))
        else:
            outs.append(mpc.run(steps, process_noise="env", seed=5, graph=graph))
    for k in ("state", "action", "cost"):
        assert np.all(np.isfinite(outs[0][k]))
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("name,kw", [("cartpole", dict(dt=0.02)), ("pendulum", dict(dt=0.05))])
def test_staged_line_search_is_bitwise_the_one_launch_line_search(oracle, name, kw):
    """Scheduling only: alphas[0] for everybody, the rest for the compacted list of rejecting problems
    (the reference's sequential first-accept loop, ilqr.py:187-202), and the two-pass form (returns of every alpha
    first, the accepted one rolled out again and written) must give the same numbers as rolling every alpha out at
    once."""
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    rng = np.random.default_rng(21)
    N, T = 300, 60
    env0 = envs.make(name, **kw)
    x0 = np.asarray(env0.x0)[None] + 0.3 * rng.standard_normal((N, env0.dm_state))
    kff0 = 1e-4 * rng.standard_normal((N, env0.dm_act, T))
    res = []
    for staged in (0, 1, 2):
        env = envs.make(name, **kw)
        s = BatchedILQR(env, T, x0, kff_init=kff0, action_penalty=1e-5, staged_line_search=staged)
        assert s.plan()["staged_line_search"] == staged
        used = []
        s.init_rollout()
        for k in (1, 1, 3, 1):   # (3: iterations chained inside one call)
            s.iterate(k)
            used.append(s.host("alpha").copy())
        res.append({k: s.host(k).copy() for k in ("xref", "uref", "K", "kff", "last_return", "lmbda", "alpha", "status")})
        res[-1]["alphas_used"] = np.stack(used)
    if name == "cartpole":  # the pendulum accepts alphas[0] throughout: it only checks the all-accept schedule
        assert len(np.unique(res[0]["alphas_used"])) > 1, "the case must exercise alphas beyond the first"
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k], equal_nan=True), k
        assert np.array_equal(res[0][k], res[2][k], equal_nan=True), k
    oe = oracle.make_env(name, dt=kw["dt"], uw=[1e-5])
    for p in range(0, N, 37):
        r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=6)
        assert abs(res[1]["last_return"][p] - r["last_return"]) <= TOL * abs(r["last_return"])


@pytest.mark.parametrize("name,gl", [("quadrotor", None), ("cartpole", 0), ("cartpole", 1)])
def test_driver_reads_dV_in_the_gains_layout(oracle, name, gl):
    """The line search's expected reduction -alpha (dV[0] + alpha dV[1]) (ilqr.py:195) must use THIS problem's dV:
    the backward kernels store dV with the gains group's layout ((N, 2) for PROBLEM_MAJOR, (2, N) for MINOR).
    min_imp > 0 makes the accept decision depend on the magnitude of `expected`, not only on its sign."""
    from trajopt_amd.ilqr import BatchedILQR
    pe, oe = _pair(name, dt=0.02)
    N, T = 7, 40
    rng = np.random.Generator(np.random.PCG64(99))
    x0 = np.asarray(pe.x0)[None] + 0.1 * rng.standard_normal((N, pe.dm_state))
    kff0 = (0.5 * 9.81 / 4 if name == "quadrotor" else 0.0) + 1e-2 * rng.standard_normal((N, pe.dm_act, T))
    s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, gains_layout=gl, min_imp=0.3)
    s.init_rollout()
    xr, ur = s.host("xref").copy(), s.host("uref").copy()
    s.iterate(1)
    dV = s._dV.cpu().numpy()
    dV = dV.T if s.gains_layout == 1 else dV          # -> (N, 2)
    for p in range(N):
        q = oracle.quadratize(oe, T, xr[p], ur[p], np.ones(T + 1))
        rb, failed = oracle.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], 1.0, 1,
                                          pe.dm_state, pe.dm_act, T)
        assert not failed
        assert_close(dV[p], rb[7], "dV[%d]" % p)
    traces = s.run(nb_iter=3)
    for p in range(N):
        r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=4, min_imp=0.3)
        assert len(traces[p]) == len(r["trace"]), (p, traces[p], r["trace"])
        assert_close(np.array(traces[p]), r["trace"], "trace[%d]" % p)
        assert s.host("alpha")[p] == pytest.approx(r["alpha"], rel=1e-14, nan_ok=True)


def test_trace_keeps_every_accepted_return_beyond_the_initial_capacity(oracle):
    """ilqr.py:219 appends every accepted return: the device trace buffer grows instead of dropping entries, across
    repeated iterate() calls and in the drop-in class with nb_iter > 100."""
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR, iLQR
    pe, oe = _pair("cartpole", dt=0.02, uw=[1e-5])
    N, T = 5, 40
    rng = np.random.Generator(np.random.PCG64(3))
    x0 = np.asarray(pe.x0)[None] + 1e-2 * rng.standard_normal((N, 4))
    kff0 = 1e-4 * rng.standard_normal((N, 1, T))
    s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, trace_cap=3)
    s.init_rollout()
    for _ in range(4):
        s.iterate(3)
    traces = s.run(nb_iter=0)
    assert s.trace_cap >= 13
    for p in range(N):
        r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=12)
        assert len(r["trace"]) > 3 and len(traces[p]) == len(r["trace"])
        assert_close(np.array(traces[p]), r["trace"], "trace[%d]" % p)
    env = envs.make("LQR-TO-v0")
    np.random.seed(5)
    kw = dict(tolfun=-np.inf, tolgrad=0.0, max_lmbda=np.inf)
    solver = iLQR(env, nb_steps=30, init_state=np.array([1.0, -0.5]), action_penalty=np.array([1e-2]), **kw)
    kff = solver.ctl.kff.copy()
    trace = solver.run(nb_iter=130)
    ref = oracle.ilqr_run(oracle.make_env("lqr", uw=[1e-2]), 30, np.array([1.0, -0.5]), kff, nb_iter=130, **kw)
    assert len(trace) == len(ref["trace"]) and len(trace) > 101
    assert_close(np.array(trace[:5]), ref["trace"][:5], "trace head")


def test_reset_takes_torch_and_numpy_kff_alike(oracle):
    """BatchedILQR.reset(kff_init=...) takes the logical (N, m, T) array whether it is NumPy or torch, in both gains
    layouts (BatchedMPC forwards kff_inits[t] unchanged)."""
    import torch
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR
    for name, gl in (("double_pendulum", 1), ("double_pendulum", 0)):
        pe = envs.make(name, dt=0.02)
        N, T, m = 6, 25, pe.dm_act
        rng = np.random.Generator(np.random.PCG64(8))
        x0 = np.asarray(pe.x0)[None] + 1e-2 * rng.standard_normal((N, pe.dm_state))
        kff0 = 1e-2 * rng.standard_normal((N, m, T))
        s = BatchedILQR(pe, T, x0, kff_init=np.zeros((N, m, T)), action_penalty=1e-5, gains_layout=gl)
        out = []
        for k in (kff0, torch.as_tensor(kff0), torch.as_tensor(kff0, device="cuda")):
            s.reset(x0, kff_init=k)
            assert np.array_equal(s.host("kff"), kff0)
            s.run(nb_iter=2)
            out.append((s.host("last_return").copy(), s.host("uref").copy()))
        for o in out[1:]:
            assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1])


def test_riccati_forward_pass_and_rollout_equal_the_stepwise_loop():
    """Riccati.forward_pass / .rollout run as one kernel launch with the noise drawn up front; the reference steps the
    env (riccati.py:52-98).  Same generator state -> the same episodes as the stepwise loop written out here with
    env.reset / env.step / env.cost (one device call each)."""
    from trajopt_amd import envs
    from trajopt_amd.riccati import Riccati
    env = envs.make("Pendulum-TO-v0")
    env.unwrapped.dt = 0.05
    T, E = 25, 3
    solver = Riccati(env, nb_steps=T, init_state=env.init())
    rng = np.random.default_rng(11)
    solver.ctl.K = 0.5 * rng.standard_normal((1, 2, T))
    solver.ctl.kff = 0.3 * rng.standard_normal((1, T))
    w = solver.weighting

    def stepwise(episodes):
        u_ = env.unwrapped
        xs, us, xns, cs = np.zeros((2, T, episodes)), np.zeros((1, T, episodes)), np.zeros((2, T, episodes)), np.zeros((T + 1, episodes))
        for ep in range(episodes):
            x = env.reset()
            for t in range(T):
                u = solver.ctl.kff[:, t] + solver.ctl.K[..., t] @ x
                us[:, t, ep] = u
                cs[t, ep] = u_.cost(x, u, us[:, t - 1, ep] if t > 0 else np.zeros(1), w[t])
                xs[:, t, ep] = x
                x, _, _, _ = env.step(u)
                xns[:, t, ep] = x
            cs[T, ep] = u_.cost(x, np.zeros(1), np.zeros(1), w[T])
        return xs, us, xns, cs
    env.seed(5)
    xs, us, xns, cs = stepwise(1)
    env.seed(5)
    state, action, cost = solver.forward_pass(solver.ctl)
    assert state.shape == (2, T + 1) and action.shape == (1, T) and cost.shape == (T + 1,)
    assert_close(state[:, :T], xs[..., 0], "state"); assert_close(state[:, 1:], xns[..., 0], "next state")
    assert_close(action, us[..., 0], "action"); assert_close(cost, cs[:, 0], "cost")
    env.seed(9)
    xs, us, xns, cs = stepwise(E)
    env.seed(9)
    data = solver.rollout(E)
    assert_close(data["x"], xs, "x"); assert_close(data["u"], us, "u"); assert_close(data["xn"], xns, "xn")
    assert_close(data["c"][-1], cs[-1], "final cost")
    assert_close(data["c"][:-1], np.repeat(cs[:-1, -1:], E, axis=1), "step costs (riccati.py:71: the last episode's row)")


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,T", [("double_cartpole", dict(dt=0.01, uw=[1e-5]), 21), ("quadrotor", dict(dt=0.02, uw=[1e-3]), 21),
                                       ("cartpole", dict(dt=0.02, uw=[1e-5]), 19), ("pendulum", dict(dt=0.05, uw=[1e-5]), 13)])
def test_two_pass_line_search_is_bitwise_the_one_launch_line_search_on_every_model(oracle, name, kw, T):
    """the two-pass line search takes the returns from the returns-only rollout and the accepted trajectory
    from ANOTHER instantiation (time-blocked, straight-line or rolled; for the quadrotor the half-block stage with
    PROBLEM_MAJOR gains - through LDS in the one-launch form, per lane in pass 2).  Forced (mode 2) on every model, with a horizon that is not a multiple of the 8-step block
    (the flush of the partial last block / half block in pass 2), every output must equal the one-launch line search bit for bit."""
    from trajopt_amd.ilqr import BatchedILQR
    pe, oe = _pair(name, **kw)
    N = 300
    rng = np.random.Generator(np.random.PCG64(77))
    x0 = np.asarray(pe.x0)[None] + 0.2 * _rand_state(name, rng, N)
    kff0 = 1e-2 * rng.standard_normal((N, pe.dm_act, T))
    keys = ("xref", "uref", "K", "kff", "lmbda", "dlmbda", "status", "iters", "last_return", "alpha")
    runs = {}
    for mode in (2, 0):
        s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, staged_line_search=mode)
        assert s.plan()["staged_line_search"] == mode
        s.run(nb_iter=3)
        traces = s.run(nb_iter=2)
        runs[mode] = dict(traces=traces, **{k: s.host(k) for k in keys})
    a, b = runs[2], runs[0]
    assert a["traces"] == b["traces"]
    for k in keys:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert max(len(t) for t in a["traces"]) > 1, "some problem must accept a step"
    p = int(np.argmax([len(t) for t in a["traces"]]))
    r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=5)
    assert abs(a["last_return"][p] - r["last_return"]) <= 1e-9 * abs(r["last_return"])


def test_mpc_with_device_drawn_kff_noise_is_reproducible_and_stays_on_the_device(oracle):
    """the per-solve initial feed-forward noise (ilqr.py:70) comes from a device generator by default -
    same seed, same run, bit for bit; a different seed differs; and the draw has the reference's scale (1e-4)."""
    import torch
    from trajopt_amd import envs
    from trajopt_amd.mpc import BatchedMPC
    N, horizon, steps = 64, 24, 3
    rng = np.random.Generator(np.random.PCG64(3))
    x0 = np.array([0., np.pi, 0., 0.])[None] + 1e-2 * rng.standard_normal((N, 4))
    runs = []
    for seed in (5, 5, 6):
        mpc = BatchedMPC(envs.make("cartpole", dt=0.02), horizon, x0, nb_iter=3, action_penalty=np.array([1e-5]))
        runs.append(mpc.run(steps, seed=seed))
    assert np.array_equal(runs[0]["action"], runs[1]["action"]) and np.array_equal(runs[0]["state"], runs[1]["state"])
    # (the third run only has to work: these cartpoles saturate at -umax whatever the 1e-4 noise, so its results may coincide)
    assert np.all(np.isfinite(runs[0]["state"]))
    s = mpc.solver
    g = torch.Generator(device=s.device)
    g.manual_seed(1)
    s.reset(torch.as_tensor(x0, device=s.device), generator=g)
    k = s.host("kff").copy()
    assert k.shape == (N, 1, horizon) and abs(k.std() / 1e-4 - 1.0) < 0.1 and abs(k.mean()) < 2e-5
    g.manual_seed(2)
    s.reset(torch.as_tensor(x0, device=s.device), generator=g)
    assert not np.array_equal(k, s.host("kff"))
    g.manual_seed(1)
    s.reset(torch.as_tensor(x0, device=s.device), generator=g)
    assert np.array_equal(k, s.host("kff"))


@pytest.mark.parametrize("name,N,T,iters,S,kw", [
    ("cartpole", 301, 21, 4, 3, dict(dt=0.02, uw=[1e-5])),          # uneven split: 101 + 100 + 100
    ("double_cartpole", 64, 30, 3, 2, dict(dt=0.01, uw=[1e-5])),
    ("quadrotor", 48, 24, 2, 4, dict(dt=0.02)),                      # PROBLEM_MAJOR gains group, lambda ladder
])
def test_sub_batches_on_streams_solve_every_problem_as_the_oracle_and_as_one_batch(oracle, name, N, T, iters, S, kw):
    """StreamedILQR (trajopt_amd/ilqr/streamed.py): the batch as S sub-batches, each a BatchedILQR on a HIP stream of its
    own - problems are independent (ilqr.py:136-237 runs one), so every problem must come out as from one batch (same
    kernels at these sizes: bit for bit) and as from the oracle's driver, in problem order, stopping rules armed."""
    from trajopt_amd.ilqr import BatchedILQR, StreamedILQR
    pe, oe = _pair(name, **kw)
    rng = np.random.Generator(np.random.PCG64(77))
    x0 = np.asarray(pe.x0)[None] + 1e-2 * rng.standard_normal((N, pe.dm_state))
    kff0 = 1e-3 * rng.standard_normal((N, pe.dm_act, T))
    one = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False)
    t_one = one.run(nb_iter=iters)
    sub = StreamedILQR(pe, T, x0, kff_init=kff0, sub_batches=S, action_penalty=None, apply_env_kwargs=False)
    # (fewer than S only if the stream probe found fewer than S - 1 side streams that run next to the caller's: side_streams())
    assert 1 <= sub.sub_batches <= S and len(sub.parts) == sub.sub_batches and sub.bounds[-1] == N
    assert sub.plan()["sub_batches"] == sub.sub_batches
    t_sub = sub.run(nb_iter=iters)
    assert t_sub == t_one
    for k in ("xref", "uref", "K", "kff", "lmbda", "dlmbda", "status", "iters", "last_return"):
        assert np.array_equal(sub.host(k), one.host(k)), k
    for p in list(range(0, N, max(1, N // 7))) + [N - 1]:
        r = oracle.ilqr_run(oe, T, x0[p], kff0[p], nb_iter=iters)
        assert len(t_sub[p]) == len(r["trace"]), (p, t_sub[p], r["trace"])
        assert_close(np.array(t_sub[p]), r["trace"], "trace[%d]" % p)
        assert_close(sub.host("xref")[p], r["xref"], "xref", tol=1e-8)
        assert_close(sub.host("uref")[p], r["uref"], "uref", tol=1e-8)
    # a second solve after reset(), without reallocating: iterate() is asynchronous, host() waits for the side streams
    x1 = x0 + 1e-2 * rng.standard_normal(x0.shape)
    sub.reset(x1, kff0)
    one.reset(x1, kff0)
    sub.init_rollout()
    one.init_rollout()
    sub.iterate(iters)
    one.iterate(iters)
    for k in ("xref", "uref", "last_return", "iters", "status"):
        assert np.array_equal(sub.host(k), one.host(k)), k


def test_sub_batches_draw_their_noise_and_hand_out_device_vectors():
    """StreamedILQR's remaining entry points: kff_init = None draws ONE (N, m, T) block for the batch (ilqr.py:70) and deals
    it out; reset(generator=...) draws per sub-batch on the device, reproducibly; vector() is the (N,) device tensor in
    problem order."""
    import torch
    from trajopt_amd import envs
    from trajopt_amd.ilqr import StreamedILQR
    N, T = 50, 16
    env = envs.make("cartpole", dt=0.02)
    rng = np.random.Generator(np.random.PCG64(9))
    x0 = np.array([0., np.pi, 0., 0.])[None] + 1e-2 * rng.standard_normal((N, 4))
    np.random.seed(123)
    want = 1e-4 * np.random.randn(N, 1, T)
    np.random.seed(123)
    s = StreamedILQR(env, T, x0, sub_batches=3, action_penalty=1e-5)
    assert np.array_equal(s.host("kff"), want)
    s.run(nb_iter=2)
    v = s.vector("last_return")
    assert v.is_cuda and v.shape == (N,) and np.array_equal(v.cpu().numpy(), s.host("last_return"))
    assert np.array_equal(s.vector("iters").cpu().numpy(), s.host("iters"))
    draws = []
    for seed in (4, 4, 5):
        g = torch.Generator(device=s.device)
        g.manual_seed(seed)
        s.reset(x0, generator=g)
        draws.append(s.host("kff").copy())
    assert np.array_equal(draws[0], draws[1]) and not np.array_equal(draws[0], draws[2])
    assert abs(draws[0].std() / 1e-4 - 1.0) < 0.15
    # the blocks of different sub-batches are different draws
    b = s.bounds
    assert not np.array_equal(draws[0][b[0]:b[0] + 10], draws[0][b[1]:b[1] + 10])
