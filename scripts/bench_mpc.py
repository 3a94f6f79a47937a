"""MPC throughput: N receding-horizon loops advancing together, as examples/ilqr/mpc/cartpole.py:18-32
does for one system - per control step a fresh horizon solve (reset, init rollout, nb_iter = 10 iLQR iterations), apply
uref[:, 0], step the plant.  Everything stays on the device (initial kff noise from a device generator).

  python scripts/bench_mpc.py [out.json]
-> control-steps/s (N systems x control steps / wall time) for 4096 and 65 536 cartpoles at horizon 500 and 60."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from trajopt_amd import _lib, envs  # noqa: E402
from trajopt_amd.mpc import BatchedMPC  # noqa: E402


def main():
    out = {"stamp": _lib.source_stamp(), "note": "cartpole dt=0.02 (examples/ilqr/mpc/cartpole.py:13), action_penalty 1e-5, nb_iter=10, "
                                                 "default stopping rules, deterministic plant, kff noise drawn on the device", "lines": []}
    rng = np.random.Generator(np.random.PCG64(1337))
    # (graph: one control step captured once as a HIP graph and replayed - BatchedMPC.run(graph=True); the capture itself is
    # inside the timed region, once per run)
    for N, horizon, steps, graph in ((4096, 60, 20, False), (4096, 60, 40, True), (512, 60, 40, False), (512, 60, 40, True),
                                     (4096, 500, 8, False), (65536, 60, 8, False), (65536, 60, 8, True), (65536, 500, 3, False)):
        env = envs.make("cartpole", dt=0.02)
        x0 = np.asarray(env.x0)[None] + rng.multivariate_normal(np.zeros(4), env.sigma0, size=N)
        mpc = BatchedMPC(env, horizon, x0, nb_iter=10, action_penalty=np.array([1e-5]))
        mpc.run(1, to_host=False)           # cold start of every kernel
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = mpc.run(steps, to_host=False, graph=graph)
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        line = {"instances": N, "horizon": horizon, "control_steps": steps, "nb_iter": 10, "hip_graph": graph, "ms_per_control_step": sec / steps * 1e3,
                "control_steps_per_s": N * steps / sec, "horizon_problem_timesteps_per_s": N * steps * 10 * horizon / sec,
                "mean_cost_last": float(r["cost"][-1].mean()), "finite": bool(torch.isfinite(r["state"]).all())}
        print(json.dumps(line), flush=True)
        out["lines"].append(line)
        del mpc, r
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
