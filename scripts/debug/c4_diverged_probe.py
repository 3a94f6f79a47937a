"""Why do some problems of the C4 bench workload fail the backward pass for every lambda?  Runs the workload until some
problem's lambda is huge, then looks at that problem's quadratisation."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import oracle as O
from trajopt_amd import envs
from trajopt_amd.ilqr import BatchedILQR
from trajopt_amd.layout import from_device
np.set_printoptions(linewidth=220, precision=4)
N, T = 4096, 500
al = np.power(10., np.linspace(0, -3, 11))[:8]
pe = envs.make("quadrotor", dt=0.02, uw=1e-3)
rng = np.random.Generator(np.random.PCG64(3001))
x0 = np.asarray(pe.x0)[None] + rng.multivariate_normal(np.zeros(12), pe.sigma0, size=N)
kff0 = 1e-4 * rng.standard_normal((N, 4, T))
s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, alphas=al,
                tolfun=-float("inf"), tolgrad=0.0, max_lmbda=float("inf"), trace_cap=40)
s.init_rollout()
prev = None
for it in range(4):
    lam0 = s.host("lmbda").copy(); ret0 = s.host("last_return").copy()
    s.iterate(1); torch.cuda.synchronize()
    lam = s.host("lmbda"); div = s._diverge.cpu().numpy(); ret = s.host("last_return"); alpha = s.host("alpha")
    bad = np.where(lam > 1e3)[0]
    print("iter", it, "problems with lambda > 1e3:", bad.size, bad[:8], "returns finite:", np.isfinite(ret).all(), "max ret", ret.max(), "median ret", np.median(ret))
    if bad.size:
        p = bad[0]
        A = from_device(s._A, s.gains_layout)[p]; B = from_device(s._B, s.gains_layout)[p]
        Cxx = from_device(s._Cxx, s.gains_layout)[p]; Cuu = from_device(s._Cuu, s.gains_layout)[p]
        xr = s.host("xref")[p]; ur = s.host("uref")[p]
        print(" problem", p, "lam0", lam0[p], "lam", lam[p], "diverge", div[p], "ret0", ret0[p], "ret", ret[p], "alpha", alpha[p])
        print(" finite: A", np.isfinite(A).all(), "B", np.isfinite(B).all(), "Cxx", np.isfinite(Cxx).all(), "xref", np.isfinite(xr).all(), "uref", np.isfinite(ur).all())
        print(" max|A|", np.nanmax(np.abs(A)), "max|B|", np.nanmax(np.abs(B)), "max|xref| per state", np.nanmax(np.abs(xr), axis=1))
        t = div[p]
        if t >= 0:
            print(" at failing step", t, "x", xr[:, t], "u", ur[:, t], "\n A", A[:, :, t], "\n B", B[:, :, t])
        break
