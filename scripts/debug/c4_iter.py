import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import oracle as O
from trajopt_amd import envs
from trajopt_amd.ilqr import BatchedILQR
np.set_printoptions(linewidth=200, precision=6)
T, D = 500, 32
al = O.DEFAULT_ALPHAS[:8]
pe = envs.make("quadrotor", dt=0.02, uw=1e-3)
oe = O.make_env("quadrotor", dt=0.02, uw=[1e-3]*4)
rng = np.random.Generator(np.random.PCG64(1337))
x0 = np.asarray(pe.x0)[None] + 1e-2 * rng.standard_normal((D, 12))
kff0 = 1e-4 * rng.standard_normal((D, 4, T))
s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, alphas=al)
s.init_rollout()
for it in range(2):
    xr, ur = s.host("xref").copy(), s.host("uref").copy()
    lam = s.host("lmbda").copy()
    s.iterate(1)
    torch.cuda.synchronize()
    cr = s._cand_ret.cpu().numpy(); dV = s._dV.cpu().numpy(); fin = s._cand_finite.cpu().numpy()
    K = s.host("K") if False else None
    from trajopt_amd.layout import from_device
    cK = from_device(s._cand_K, s.gains_layout); ckff = from_device(s._cand_kff, s.gains_layout)
    p = 0
    print("iter", it, "lam", lam[p], "->", s.host("lmbda")[p], "alpha", s.host("alpha")[p], "ret", s.host("last_return")[p], "diverge", s._diverge.cpu().numpy()[p])
    print(" gpu cand_ret", cr[:, p], "dV", dV[:, p], "fin", fin[:, p])
    q = O.quadratize(oe, T, xr[p], ur[p], np.ones(T + 1))
    rb, failed = O.backward_pass(q["Cxx"], q["cx"], q["Cuu"], q["cu"], q["Cxu"], q["A"], q["B"], lam[p], 1, 12, 4, T)
    print(" oracle failed", failed, "dV", rb[7], "K err", np.abs(rb[8]-cK[p]).max()/np.abs(rb[8]).max(), "kff err", np.abs(rb[9]-ckff[p]).max()/np.abs(rb[9]).max())
    rets = []
    for a in al:
        st, ac, co = O.forward_pass(oe, T, x0[p], cK[p], ckff[p], a, xr[p], ur[p], np.ones(T + 1))
        rets.append(co.sum())
    print(" oracle rets (gpu gains)", np.array(rets))
    print(" xref max", np.abs(s.host("xref")[p]).max(axis=1))
r = O.ilqr_run(oe, T, x0[0], kff0[0], alphas=al, nb_iter=2)
print("oracle trace", r["trace"], "lmbda", r["lmbda"], "alpha", r.get("alpha"))
