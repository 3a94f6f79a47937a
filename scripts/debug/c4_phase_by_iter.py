"""Phase times of EACH iteration of the C4 bench workload (4096 quadrotors, 8 alphas): which phase makes the early
iterations long?  python scripts/debug/c4_phase_by_iter.py [iters] [max_lmbda]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from trajopt_amd import _lib, envs
from trajopt_amd.ilqr import BatchedILQR

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 14
MAXL = float(sys.argv[2]) if len(sys.argv) > 2 else 1e6
PHASES = ("prep", "quadratize", "backward", "post_backward", "rollout", "select_commit")
N, T = 4096, 500
al = np.power(10., np.linspace(0, -3, 11))[:8]
pe = envs.make("quadrotor", dt=0.02, uw=1e-3)
rng = np.random.Generator(np.random.PCG64(3001))
x0 = np.asarray(pe.x0)[None] + rng.multivariate_normal(np.zeros(12), pe.sigma0, size=N)
kff0 = 1e-4 * rng.standard_normal((N, 4, T))
s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, alphas=al,
                tolfun=-float("inf"), tolgrad=0.0, max_lmbda=MAXL, trace_cap=ITERS + 8)
s.init_rollout()
L = _lib.lib()
for it in range(ITERS):
    lam0 = s.host("lmbda").copy()
    _lib.check(L.trajopt_profile_begin(1))
    s.iterate(1)
    torch.cuda.synchronize()
    ms = (C.c_double * len(PHASES))()
    nrec = C.c_int(0)
    _lib.check(L.trajopt_profile_end(ms, C.byref(nrec)))
    st = s.host("status")
    lam1 = s.host("lmbda")
    div = s._diverge.cpu().numpy()
    print("iter %2d  %s  active %d  lam0 med %.3g  lam1 med %.3g  diverged %d" % (
        it, "  ".join("%s %.2f" % (k[:5], ms[i]) for i, k in enumerate(PHASES)), int((st == 0).sum()),
        np.median(lam0), np.median(lam1), int((div >= 0).sum())))
