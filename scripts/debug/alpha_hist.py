import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from trajopt_amd import envs
from trajopt_amd.ilqr import BatchedILQR
from bench import make_problem_set
for name, dt in (("cartpole", 0.02), ("double_cartpole", 0.01)):
    env = envs.make(name, dt=dt, uw=1e-5)
    N, T = 8192, 500
    x0, kff0 = make_problem_set(env, N, T, 1337)
    s = BatchedILQR(env, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False,
                    tolfun=-float("inf"), tolgrad=0.0, max_lmbda=float("inf"))
    s.init_rollout()
    for it in range(30):
        s.iterate(1)
        torch.cuda.synchronize()
        sel = s._sel.cpu().numpy()
        h = np.bincount(sel + 1, minlength=12)
        print(name, it, "none=%d" % h[0], "accepted per alpha:", h[1:].tolist(), flush=True)
