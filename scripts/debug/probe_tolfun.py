"""Developer aid: status counts of the 16 384-cartpole test batch for a few tolfun values."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.test_ilqr_gpu import _pair
from trajopt_amd.ilqr import BatchedILQR
pe, oe = _pair("cartpole", dt=0.02, uw=[1e-5])
N, T = 16384, 24
rng = np.random.Generator(np.random.PCG64(12))
x0 = np.asarray(pe.x0)[None] + 0.3 * rng.standard_normal((N, pe.dm_state))
kff0 = 1e-2 * rng.standard_normal((N, pe.dm_act, T))
for tolfun in (1e-6, 1e-2, 1.0, 10.0, 100.0):
    s = BatchedILQR(pe, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False, tolfun=tolfun)
    s.run(nb_iter=6)
    print(tolfun, np.bincount(s.host("status"), minlength=4).tolist(), np.bincount(s.host("iters")).tolist(), float(s.host("last_return").mean()))
