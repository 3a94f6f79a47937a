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
GPS = sorted(glob.glob(os.path.join(HERE, "gps_*.npz")))
BSP = sorted(glob.glob(os.path.join(HERE, "bsp_*.npz")))
GPS_ARGS = ("Cxx", "cx", "Cuu", "cu", "Cxu", "c0", "A", "B", "c", "sigma_dyn", "alpha")
GPS_OUT = ("Qxx", "Qux", "Quu", "qx", "qu", "q0", "V", "v", "v0", "K", "kff", "sigma_ctl")
AG = ("Cxx", "cx", "Cuu", "cu", "Cxu", "c0")
FW = ("mu_x", "sigma_x", "mu_u", "sigma_u", "mu_xu", "sigma_xu")
BSP_ARGS = ("Q", "q", "R", "r", "P", "p", "F", "G", "T", "U", "V", "X", "Y", "Z")
BSP_OUT = ("S", "s", "tau", "dS", "K", "kff")
ids = lambda fs: [os.path.basename(f)[:-4] for f in fs]


def test_fixture_set_is_complete():
    assert len(BWD) == 5 and len(RIC) == 2 and len(ILQR) == 2 and len(GPS) == 2 and len(BSP) == 2


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


# ---- SURVEY 8f.2 / 8f.4 fixtures: one MBGPS dual evaluation; the belief-space backward pass -------------------------
def _gps_chain(g, mod):
    """The five functions of gps/src/util.cpp chained as gps/mbgps.py:173-195 does, through `mod`'s namesakes."""
    n, m, T = int(g["n"]), int(g["m"]), int(g["T"])
    last = (g["last_K"], g["last_kff"], g["last_sigma"])
    ag = mod["augment_cost"](*[g["in_" + k] for k in AG], *last, g["in_alpha"], n, m, T)
    bw = mod["backward_pass"](*ag, g["in_A"], g["in_B"], g["in_c"], g["in_sigma_dyn"], g["in_alpha"], n, m, T)
    fw = mod["forward_pass"](g["mu0"], g["sigma0"], g["in_A"], g["in_B"], g["in_c"], g["in_sigma_dyn"], bw[9], bw[10],
                             bw[11], n, m, T)
    dual = mod["quad_expectation"](g["mu0"], g["sigma0"], bw[6][..., 0], bw[7][:, 0], bw[8][0])
    kl = mod["kl_divergence"](bw[9], bw[10], bw[11], *last, fw[0], fw[1], n, m, T)
    return ag, bw, fw, dual, kl


def _check_gps(g, ag, bw, fw, dual, kl, tol):
    for k, a in zip(AG, ag):
        assert_close(a, g["ag_" + k], "ag_" + k, tol=tol)
    for k, a in zip(GPS_OUT, bw[:12]):
        assert_close(a, g["bw_" + k], "bw_" + k, tol=tol)
    for k, a in zip(FW, fw):
        assert_close(a, g["fw_" + k], "fw_" + k, tol=tol)
    assert abs(dual - float(g["dual"])) <= tol * abs(float(g["dual"]))
    assert_close(kl, g["kl"], "kl", tol=10 * tol)   # differences of nearly equal gains


@pytest.mark.parametrize("path", GPS, ids=ids(GPS))
def test_oracle_reproduces_gps_fixture(oracle, path):
    g = np.load(path)
    mod = dict(augment_cost=oracle.gps_augment_cost, forward_pass=oracle.gps_forward_pass,
               backward_pass=lambda *a: oracle.gps_backward_pass(*a)[0], quad_expectation=oracle.gps_quad_expectation,
               kl_divergence=oracle.gps_kl_divergence)
    _check_gps(g, *_gps_chain(g, mod), tol=1e-13)


@pytest.mark.parametrize("path", BSP, ids=ids(BSP))
def test_oracle_reproduces_bsp_fixture(oracle, path):
    g = np.load(path)
    b, m, T = int(g["b"]), int(g["m"]), int(g["T"])
    out, failed = oracle.bsp_backward_pass(*[g["in_" + k] for k in BSP_ARGS], float(g["lmbda"]), int(g["reg"]), b, m, T)
    assert not failed
    for k, a in zip(BSP_OUT, out[:6]):
        assert_close(a, g["out_" + k], k, tol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GPS, ids=ids(GPS))
def test_hip_gps_module_reproduces_fixture(path):
    from trajopt_amd.gps import core
    g = np.load(path)
    mod = dict(augment_cost=core.augment_cost, forward_pass=core.forward_pass, backward_pass=core.backward_pass,
               quad_expectation=core.quad_expectation, kl_divergence=lambda *a: core.kl_divergence(*a)[0])
    _check_gps(g, *_gps_chain(g, mod), tol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("path", BSP, ids=ids(BSP))
def test_hip_bsp_backward_reproduces_fixture(path):
    from trajopt_amd.bspilqr.core import backward_pass
    g = np.load(path)
    b, m, T = int(g["b"]), int(g["m"]), int(g["T"])
    out = backward_pass(*[g["in_" + k] for k in BSP_ARGS], float(g["lmbda"]), int(g["reg"]), b, m, T)
    assert out[6] == 0
    for k, a in zip(BSP_OUT, out[:6]):
        assert_close(a, g["out_" + k], k)
