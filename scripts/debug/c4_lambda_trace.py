"""How many lambda-retry sweeps does the backward pass of the C4 bench workload take, per problem and iteration?
Replays each iteration's backward pass (same quadratisation, same lambda / dlambda going in) through the batched entry
and derives the retries from dlambda (ilqr.py:166-167).  python scripts/debug/c4_lambda_trace.py [N] [iters] [nalphas] [max_lmbda]"""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from trajopt_amd import backward, envs
from trajopt_amd._lib import LambdaSchedule
from trajopt_amd.ilqr import BatchedILQR

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 14
NA = int(sys.argv[3]) if len(sys.argv) > 3 else 8
MAXL = float(sys.argv[4]) if len(sys.argv) > 4 else float("inf")
T = 500
al = np.power(10., np.linspace(0, -3, 11))[:NA]
pe = envs.make("quadrotor", dt=0.02, uw=1e-3)
rng = np.random.Generator(np.random.PCG64(3001))
x0 = np.asarray(pe.x0)[None] + rng.multivariate_normal(np.zeros(12), pe.sigma0, size=N)
kff0 = 1e-4 * rng.standard_normal((N, 4, T))
s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, alphas=al,
                tolfun=-float("inf"), tolgrad=0.0, max_lmbda=MAXL, trace_cap=ITERS + 8)
s.init_rollout()
sched = LambdaSchedule(1e-6, MAXL, 1.6)
for it in range(ITERS):
    lam0, dl0 = s.lmbda.clone(), s.dlmbda.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s.iterate(1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    # replay the backward pass on this iteration's quadratisation
    lam, dl = lam0.clone(), dl0.clone()
    o = backward.ilqr_backward_pass(s._Cxx, s._cx, s._Cuu, s._cu, s._Cxu, s._A, s._B, lam, 1, 12, 4, T, N, s.gains_layout,
                                    full=False, retry=True, sched=sched, dlmbda=dl)
    torch.cuda.synchronize()
    lam0h, dl0h, lamh, dlh = (x.cpu().numpy() for x in (lam0, dl0, lam, dl))
    # k retries: dl = max(dl0 * 1.6, 1.6) then * 1.6^(k-1)
    first = np.maximum(dl0h * 1.6, 1.6)
    k = np.where(dlh == dl0h, 0, 1 + np.rint(np.log(np.maximum(dlh, 1e-300) / first) / np.log(1.6))).astype(int)
    div = o["diverge"].cpu().numpy()
    alpha = s.host("alpha")
    acc = s.host("trace_len")
    print("iter %2d  %.2f ms  lam0 [min %.3g med %.3g max %.3g]  retries: mean %.2f max %d hist %s  still-diverged %d  lam_after med %.3g"
          % (it, ms, lam0h.min(), np.median(lam0h), lam0h.max(), k.mean(), k.max(), np.bincount(k, minlength=8)[:12].tolist(),
             int((div >= 0).sum()), np.median(lamh)))
