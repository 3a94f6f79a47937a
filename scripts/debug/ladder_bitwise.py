"""Checksum of a C4-like solve (512 quadrotors, T = 200, 8 alphas, 8 iterations, max_lmbda = 1e6): run once as is and once
with TRAJOPT_NO_LADDER=1 - the three-launch lambda ladder must reproduce the in-wave loop bit for bit."""
import hashlib
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
from trajopt_amd import envs
from trajopt_amd.ilqr import BatchedILQR

N, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 200)
al = np.power(10., np.linspace(0, -3, 11))[:8]
pe = envs.make("quadrotor", dt=0.02, uw=1e-3)
rng = np.random.Generator(np.random.PCG64(3001))
x0 = np.asarray(pe.x0)[None] + rng.multivariate_normal(np.zeros(12), pe.sigma0, size=N)
kff0 = 1e-4 * rng.standard_normal((N, 4, T))
s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, alphas=al,
                tolfun=-float("inf"), tolgrad=0.0, max_lmbda=1e6)
s.init_rollout()
s.iterate(8)
h = hashlib.sha256()
for k in ("xref", "uref", "K", "kff", "lmbda", "dlmbda", "status", "iters", "last_return"):
    h.update(np.ascontiguousarray(s.host(k)).tobytes())
st = s.host("status")
print(h.hexdigest()[:16], "status counts", np.bincount(st, minlength=4).tolist(), "lam max %.3g" % s.host("lmbda").max())
