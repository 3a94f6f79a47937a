"""Host-side logic that needs no GPU: layouts, weighting schedules, env parameter structs, shard bounds."""
import numpy as np
import pytest


def test_layout_round_trip_and_memory_order():
    from trajopt_amd import layout as L
    rng = np.random.default_rng(0)
    a = rng.standard_normal((5, 3, 2, 7))   # (N, rows, cols, T)
    for lay in (0, 1):
        t = L.to_device(a, lay, device="cpu")
        assert tuple(t.shape) == L.device_shape(a.shape, lay) and t.is_contiguous()
        np.testing.assert_array_equal(L.from_device(t, lay), a)
    flat = L.to_device(a, 0, device="cpu").reshape(-1).numpy()
    p, i, j, t_ = 3, 2, 1, 4   # X(p,i,j,t) = X[p*E + i + rows*j + rows*cols*t]   (include/trajopt_amd.h)
    assert flat[p * 3 * 2 * 7 + i + 3 * j + 6 * t_] == a[p, i, j, t_]
    flat = L.to_device(a, 1, device="cpu").reshape(-1).numpy()
    assert flat[(i + 3 * j + 6 * t_) * 5 + p] == a[p, i, j, t_]
    b = rng.standard_normal((4, 3, 6))      # (N, rows, T)
    for lay in (0, 1):
        np.testing.assert_array_equal(L.from_device(L.to_device(b, lay, device="cpu"), lay), b)


def test_synthetic_inputs_follow_the_survey_recipe():
    from trajopt_amd import synthetic as S
    s = S.backward_inputs(4, 2, 6, 3)
    assert s["A"].shape == (3, 4, 4, 6) and s["Cxx"].shape == (3, 4, 4, 7) and s["Cxu"].shape == (3, 4, 2, 7)
    q = S.problem(s, 1)
    assert all(v.flags.f_contiguous for v in q.values())
    for t in range(7):
        assert np.all(np.linalg.eigvalsh(q["Cxx"][..., t]) > 0) and np.all(np.linalg.eigvalsh(q["Cuu"][..., t]) > 0)
    s2 = S.backward_inputs(4, 2, 6, 3)
    assert all(np.array_equal(s[k], s2[k]) for k in s)   # PCG64(1337) by default


def test_weighting_schedules_match_the_reference_rules():
    from trajopt_amd.ilqr.ilqr import make_weighting
    assert np.array_equal(make_weighting(5, None), np.ones(6))
    w = make_weighting(10, {"mult": 0.5, "shift": 4.0})
    t = np.linspace(0, 10, 11)
    np.testing.assert_allclose(w, 1. / (1. + np.exp(-0.5 * (t - 4.0))))
    w = make_weighting(4, {"discount": 0.9})
    np.testing.assert_allclose(w, [1, 0.9, 0.81, 0.729, 0.6561])
    with pytest.raises(NotImplementedError):
        make_weighting(4, {"foo": 1})


def test_env_registry_and_struct():
    from oracle import oracle as O
    from trajopt_amd import envs
    for name, ref_id in (("pendulum", "Pendulum-TO-v0"), ("cartpole", "Cartpole-TO-v0"),
                         ("double_cartpole", "DoubleCartpole-TO-v0"), ("lqr", "LQR-TO-v0"),
                         ("double_pendulum", "DoublePendulum-TO-v0"), ("robot", "Robot-TO-v0"), ("quad_pendulum", "QuadPendulum-TO-v0"),
                         ("lqr_v1", "LQR-TO-v1"), ("lqr_v2", "LQR-TO-v2")):
        e = envs.make(ref_id)
        assert type(e) is type(envs.make(name))
        assert e.unwrapped is e and e.action_space.shape == (e.dm_act,) and e.observation_space.shape == (e.dm_state,)
        assert bytes(e.struct()) == bytes(O.make_env(name))
    for name, ref_id in (("pendulum", "Pendulum-TO-v1"), ("cartpole", "Cartpole-TO-v1"),
                         ("double_cartpole", "DoubleCartpole-TO-v1"), ("double_pendulum", "DoublePendulum-TO-v1"),
                         ("quad_pendulum", "QuadPendulum-TO-v1")):
        e = envs.make(ref_id)                  # the ...WithCartesianCost classes
        assert type(e) is type(envs.make(name, cartesian=True)) and e.features is not None
        assert bytes(e.struct()) == bytes(O.make_env(name, cartesian=True))
        assert e.struct().nb_features == len(e.g) == len(e.gw)
    e = envs.make("Cartpole-TO-v0")
    e.unwrapped.dt = 0.02                      # the reference's scripts mutate attributes after make()
    e.unwrapped.uw = 1e-5 * np.ones(1)
    e.unwrapped.slew_rate = True
    s = e.struct()
    assert s.dt == 0.02 and s.slew_rate == 1 and s.uw[0] == 1e-5 and s.umax[0] == 10.0 and s.xmax[0] == 10.0
    assert np.isinf(s.xmax[1]) and s.periodic == 0
    e.periodic = True
    assert e.struct().periodic == 0b0010
    x0, sig = e.init()
    assert np.array_equal(x0, [0., np.pi, 0., 0.]) and sig.shape == (4, 4)
    with pytest.raises(KeyError):
        envs.make("Car-TO-v0")                 # out of scope
    lq = envs.LQRv0(A=np.eye(3), B=np.ones((3, 2)), g=np.zeros(3))
    s = lq.struct()
    assert (s.dm_state, s.dm_act, s.cost_times_dt) == (3, 2, 1) and s.lin_B[4] == 1.0 and np.isinf(s.umax[1])


def test_shard_bounds_cover_the_problem_axis_exactly():
    from trajopt_amd.sharding import shard, shard_bounds
    for N in (1, 7, 8, 65536, 65537):
        for G in (1, 2, 3, 8):
            spans = [shard_bounds(N, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            assert all(spans[r][1] == spans[r + 1][0] for r in range(G - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(65536, 8, 3) == (3 * 8192, 4 * 8192)     # config C3: 8192 problems per GPU
    a = np.arange(10)
    assert np.array_equal(np.concatenate([shard(a, 3, r) for r in range(3)]), a)
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_reference_containers_keep_names_and_shapes():
    from trajopt_amd.ilqr import objects as O
    from trajopt_amd.riccati import objects as R
    lc = O.LinearControl(4, 2, 9)
    assert lc.K.shape == (2, 4, 9) and lc.kff.shape == (2, 9)
    x, xref, uref = np.ones((4, 10)), np.zeros((4, 10)), np.zeros((2, 9))
    lc.K[..., 3] = 0.5
    lc.kff[..., 3] = 2.0
    np.testing.assert_allclose(lc.action(x, 0.25, xref, uref, 3), 0.5 + 2.0)   # objects.py:137-139
    qf = O.QuadraticStateActionValue(4, 2, 9)
    assert qf.Qux.shape == (2, 4, 9) and qf.Quu.shape == (2, 2, 9) and O.QuadraticStateValue(4, 10).V.shape == (4, 4, 10)
    c = O.QuadraticCost(4, 2, 10)
    assert len(c.params) == 5 and c.Cxu.shape == (4, 2, 10)
    d = R.LinearDynamics(4, 2, 9)
    assert len(d.params) == 3 and d.c.shape == (4, 9)
    rc = R.LinearControl(4, 2, 9)
    rc.K[..., 0], rc.kff[..., 0] = 1.0, 3.0
    np.testing.assert_allclose(rc.action(np.ones(4), 0), 3.0 + 4.0)            # riccati/objects.py:128-129
