"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle after it was
cross-checked by the NumPy mirror and the long-double build; the reference itself holds no vectors).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the C ABI."""
import glob
import os

import numpy as np
import pytest

from tests.util import assert_close

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT = ("Qxx", "Qux", "Quu", "qx", "qu", "V", "v", "dV", "K", "kff")
KEYS = ("Cxx", "cx", "Cuu", "cu", "Cxu", "A", "B")
BWD = sorted(glob.glob(os.path.join(HERE, "bwd_*.npz")))
RIC = sorted(glob.glob(os.path.join(HERE, "ric_*.npz")))
ILQR = sorted(glob.glob(os.path.join(HERE, "ilqr_*.npz")))
ids = lambda fs: [os.path.basename(f)[:-4] for f in fs]


def test_fixture_set_is_complete():
    assert len(BWD) == 5 and len(RIC) == 2 and len(ILQR) == 2


@pytest.mark.parametrize("path", BWD, ids=ids(BWD))
def test_oracle_reproduces_backward_fixture(oracle, path):
    g = np.load(path)
    n, m, T = int(g["n"]), int(g["m"]), int(g["T"])
    out, failed = oracle.backward_pass(*[g["in_" + k] for k in KEYS], float(g["lmbda"]), int(g["reg"]), n, m, T)
    assert not failed
    for k, a in zip(OUT, out[:10]):
        assert_close(a, g["out_" + k], k, tol=1e-13)


@pytest.mark.parametrize("path", RIC, ids=ids(RIC))
def test_oracle_reproduces_riccati_fixture(oracle, path):
    g = np.load(path)
    n, m, T = int(g["n"]), int(g["m"]), int(g["T"])
    got = oracle.riccati_backward_pass(*[g["in_" + k] for k in KEYS], g["in_c"], n, m, T)
    for k, a in zip(("K", "kff", "V", "v"), got):
        assert_close(a, g["out_" + k], k, tol=1e-13)


def _env_of(path, mod):
    g = np.load(path)
    name = "pendulum" if "pendulum" in path else "cartpole"
    return g, name, float(g["dt"]), [float(x) for x in g["uw"]]


@pytest.mark.parametrize("path", ILQR, ids=ids(ILQR))
def test_oracle_reproduces_ilqr_fixture(oracle, path):
    g, name, dt, uw = _env_of(path, oracle)
    e = oracle.make_env(name, dt=dt, uw=uw)
    T = int(g["T"])
    r = oracle.ilqr_run(e, T, g["x0"], g["kff0"], nb_iter=int(g["iters"]))
    assert_close(r["trace"], g["trace"], "trace", tol=1e-13)
    assert_close(r["xref"], g["xref"], "xref", tol=1e-12)
    assert_close(r["K"], g["K"], "K", tol=1e-11)
    assert r["status"] == int(g["status"])


# ---- the HIP path against the same fixtures --------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("path", BWD, ids=ids(BWD))
def test_hip_backward_reproduces_fixture(path):
    from trajopt_amd.ilqr.core import backward_pass
    g = np.load(path)
    n, m, T = int(g["n"]), int(g["m"]), int(g["T"])
    out = backward_pass(*[g["in_" + k] for k in KEYS], float(g["lmbda"]), int(g["reg"]), n, m, T)
    assert out[10] == 0
    for k, a in zip(OUT, out[:10]):
        assert_close(a, g["out_" + k], k)   # 1e-9 relative, north_star


@pytest.mark.gpu
@pytest.mark.parametrize("path", RIC, ids=ids(RIC))
def test_hip_riccati_reproduces_fixture(path):
    from trajopt_amd.riccati.riccati import backward_pass_host
    g = np.load(path)
    n, m, T = int(g["n"]), int(g["m"]), int(g["T"])
    got = backward_pass_host(*[g["in_" + k] for k in KEYS], g["in_c"], n, m, T)
    for k, a in zip(("K", "kff", "V", "v"), got):
        assert_close(a, g["out_" + k], k)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ILQR, ids=ids(ILQR))
def test_hip_ilqr_reproduces_fixture(path):
    from trajopt_amd import envs
    from trajopt_amd.ilqr import iLQR
    from trajopt_amd.ilqr.objects import LinearControl, quadratize
    g, name, dt, uw = _env_of(path, None)
    env = envs.make(name, dt=dt)
    T = int(g["T"])
    s = iLQR(env, T, g["x0"], action_penalty=np.asarray(uw))
    s.ctl.kff = np.array(g["kff0"])
    trace = s.run(nb_iter=int(g["iters"]))
    assert_close(np.array(trace), g["trace"], "trace")          # trajectory cost, 1e-9
    assert_close(s.xref, g["xref"], "xref", tol=1e-8)
    assert_close(s.uref, g["uref"], "uref", tol=1e-8)
    assert_close(s.ctl.K, g["K"], "K", tol=1e-8)
    assert_close(s.vfunc.V, g["V"], "V", tol=1e-8)
    assert s.status == int(g["status"]) and s.lmbda == pytest.approx(float(g["lmbda"]), rel=1e-14)
    # kernels in isolation on the fixture's trajectory
    q = quadratize(env, T, g["xref"], g["uref"], np.ones(T + 1))
    for k in ("A", "B", "Cxx", "cx", "Cuu", "cu", "Cxu"):
        den = max(1.0, np.abs(g["q_" + k]).max())
        assert np.abs(q[k] - g["q_" + k]).max() <= 1e-10 * den, k
    lc = LinearControl(env.dm_state, env.dm_act, T)
    lc.K, lc.kff = np.array(g["K"]), np.array(g["kff"])
    s.xref, s.uref = np.array(g["xref"]), np.array(g["uref"])
    st, ac, co = s.forward_pass(lc, float(g["fp_alpha"]))
    assert_close(st, g["fp_state"], "state"); assert_close(ac, g["fp_action"], "action")
    assert_close(co, g["fp_cost"], "cost")
