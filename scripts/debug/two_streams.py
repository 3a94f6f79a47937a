"""Developer aid: one batch of NP cartpoles on one stream against two half batches on two streams (problems are
independent, so the halves are two solvers; the latency-bound launches of one may run under the other's rollout)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from trajopt_amd import envs
from trajopt_amd.ilqr import BatchedILQR
from bench import make_problem_set
env = envs.make("cartpole", dt=0.02, uw=1e-5)
N, T = int(os.environ.get("NP", 65536)), 500
K = int(os.environ.get("ITERS", 8))
x0, kff0 = make_problem_set(env, N, T, 1337)
kw = dict(action_penalty=None, apply_env_kwargs=False, tolfun=-float("inf"), tolgrad=0.0, max_lmbda=float("inf"))

def run(parts, chain):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    h = N // parts
    solvers = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            s = BatchedILQR(env, T, x0[i * h:(i + 1) * h], kff_init=kff0[i * h:(i + 1) * h], **kw)
            s.init_rollout(); s.iterate(2)
            solvers.append(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if chain:
        for s, st in zip(solvers, streams):
            with torch.cuda.stream(st):
                s.iterate(K)
    else:
        for _ in range(K):
            for s, st in zip(solvers, streams):
                with torch.cuda.stream(st):
                    s.iterate(1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    ret = np.concatenate([s.host("last_return") for s in solvers])
    print("parts", parts, "chained" if chain else "interleaved", "ms/iteration %.3f" % dt, "mean return %.9f" % ret.mean(),
          "plan", solvers[0].plan()["staged_line_search"])

for parts, chain in ((1, True), (2, True), (2, False), (4, True)):
    run(parts, chain)
