"""Experiment: one batch of N problems against S sub-batches of N / S problems, each on a HIP stream of its own
(problems are independent: ilqr.py:136-237 runs one problem at a time).  Prints ms per iLQR iteration of the whole set."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from trajopt_amd import envs
from trajopt_amd.ilqr import BatchedILQR


def run(env_name, N, S, T=500, nb_alphas=11, K=20, W=4, max_lmbda=float("inf")):
    alphas = bench.ALL_ALPHAS[:nb_alphas]
    dt, uw = bench.ENV_DEFAULTS[env_name]
    env = envs.make(env_name, dt=dt, uw=uw)
    x0, kff0 = bench.make_problem_set(env, N, T, 100)
    streams = [torch.cuda.Stream() for _ in range(S)]
    solvers = []
    per = N // S
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            s = BatchedILQR(env, T, x0[i * per:(i + 1) * per], kff_init=kff0[i * per:(i + 1) * per], action_penalty=None,
                            apply_env_kwargs=False, alphas=alphas, tolfun=-float("inf"), tolgrad=0.0, max_lmbda=max_lmbda,
                            trace_cap=256)
            s.init_rollout()
            solvers.append(s)
    torch.cuda.synchronize()

    def go(k):
        for s, st in zip(solvers, streams):
            with torch.cuda.stream(st):
                s.iterate(k)
    go(W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(K)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / K
    it = sum(int(s.host("iters").astype(np.int64).sum()) for s in solvers)
    print(env_name, "N", N, "max_lmbda", max_lmbda, "sub-batches", S, "ms/iteration", round(ms, 4), "problem-iterations", it, flush=True)
    del solvers
    torch.cuda.empty_cache()


def run_cls(env_name, N, S, seed, T=500, nb_alphas=8, K=8, W=3, max_lmbda=1e6):
    """the same measurement through StreamedILQR (S = 1: BatchedILQR), problem set of `seed`"""
    from trajopt_amd.ilqr import StreamedILQR
    alphas = bench.ALL_ALPHAS[:nb_alphas]
    dt, uw = bench.ENV_DEFAULTS[env_name]
    env = envs.make(env_name, dt=dt, uw=uw)
    x0, kff0 = bench.make_problem_set(env, N, T, seed)
    kw = dict(action_penalty=None, apply_env_kwargs=False, alphas=alphas, tolfun=-float("inf"), tolgrad=0.0, max_lmbda=max_lmbda,
              trace_cap=256)
    s = StreamedILQR(env, T, x0, kff_init=kff0, sub_batches=S, **kw) if S > 1 else BatchedILQR(env, T, x0, kff_init=kff0, **kw)
    s.init_rollout()
    s.iterate(W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.iterate(K)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / K
    print("class flow", env_name, "N", N, "seed", seed, "sub-batches", S, "ms/iteration", round(ms, 4), "problem-iterations",
          int(s.host("iters").astype(np.int64).sum()), flush=True)
    del s
    torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        inf = float("inf")
        for env_name, N in (("cartpole", 65536), ("cartpole", 32768), ("cartpole", 16384), ("cartpole", 8192), ("pendulum", 65536),
                            ("double_cartpole", 4096), ("double_cartpole", 6144), ("double_cartpole", 12288), ("double_cartpole", 32768),
                            ("double_cartpole", 65536)):
            for S in (1, 2, 3, 4):
                run_cls(env_name, N - N % S, S, 2001, nb_alphas=11, K=10, W=3, max_lmbda=inf)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c4cls":
        for seed in (100, 1337, 3001, 100, 1337):
            for S in (4, 3, 2, 1):
                run_cls("quadrotor", 4096 - 4096 % S, S, seed)
        for S in (1, 2, 3, 4):
            run_cls("quadrotor", 4096 - 4096 % S, S, 3001, max_lmbda=float("inf"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c4":
        for rep in range(3):
            for S in (1, 2, 4):
                run("quadrotor", 4096, S, nb_alphas=8, max_lmbda=1e6, K=8, W=3)
        for S in (1, 2, 4):
            run("quadrotor", 4096, S, nb_alphas=8, max_lmbda=1e6, K=20, W=11)
        sys.exit(0)
    cases = [("quadrotor", 4096, 8, 1e6, (1, 2, 4, 8)), ("quadrotor", 4096, 8, float("inf"), (1, 2, 4, 8)),
             ("double_cartpole", 8192, 11, float("inf"), (1, 2, 4)), ("double_cartpole", 16384, 11, float("inf"), (1, 2, 4)),
             ("cartpole", 4096, 11, float("inf"), (1, 2)), ("cartpole", 16384, 11, float("inf"), (1, 2)),
             ("cartpole", 32768, 11, float("inf"), (1, 2, 4)), ("cartpole", 65536, 11, float("inf"), (1, 2, 4))]
    for env_name, N, na, ml, Ss in cases:
        for S in Ss:
            run(env_name, N - N % S, S, nb_alphas=na, max_lmbda=ml, K=10 if N >= 65536 else 20)
