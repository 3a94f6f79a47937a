"""Developer aid: device time of the backward phase of the bench workload (no result checks)."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from trajopt_amd import _lib, envs
from trajopt_amd.ilqr import BatchedILQR
from bench import make_problem_set, PHASES
env = envs.make("cartpole", dt=0.02, uw=1e-5)
N, T = 4096, 500
x0, kff0 = make_problem_set(env, N, T, 1337)
s = BatchedILQR(env, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, tolfun=-float("inf"), tolgrad=0.0,
                max_lmbda=float("inf"))
L = _lib.lib()
s.init_rollout(); s.iterate(3); torch.cuda.synchronize()
_lib.check(L.trajopt_profile_begin(10)); s.iterate(10); torch.cuda.synchronize()
ms = (C.c_double * 6)(); n = C.c_int(0)
_lib.check(L.trajopt_profile_end(ms, C.byref(n)))
print({k: round(ms[i] / n.value, 4) for i, k in enumerate(PHASES)})
if os.environ.get("TRAJOPT_AMD_SO", "").endswith("diag3.so"):
    s.iterate(1); torch.cuda.synchronize()
    dV = s._dV.cpu().numpy()
    print("cycles per wave: expansion mean %.0f  recursion mean %.0f   (min/max exp %.0f/%.0f, rec %.0f/%.0f)" % (
        dV[0].mean(), dV[1].mean(), dV[0].min(), dV[0].max(), dV[1].min(), dV[1].max()))
if os.environ.get("TRAJOPT_AMD_SO", "").endswith("diag4.so"):
    for it in range(8):
        s.iterate(1); torch.cuda.synchronize()
        g = s._g_norm.cpu().numpy()
        print("sweeps per problem histogram (0 = one sweep):", np.bincount(g.astype(int)).tolist())
