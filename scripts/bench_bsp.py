"""Time the belief-space iLQR backward pass on resident inputs: ms per launch and algorithmic GB/s."""
import json
import sys

import numpy as np
import torch

from trajopt_amd import backward, layout as L, synthetic as S

ARGS = ("Q", "q", "R", "r", "P", "p", "F", "G", "T", "U", "V", "X", "Y", "Z")


def main():
    out = []
    for (b, m, T, N) in [(2, 2, 100, 4096), (2, 2, 100, 65536), (4, 2, 100, 4096), (4, 2, 100, 4160), (4, 2, 100, 32768), (4, 2, 100, 32832),
                         # no lane-per-problem instantiation: wave = problem with run-time dimensions (gen_wave.hip)
                         (6, 2, 100, 1024), (6, 2, 100, 4096), (3, 1, 100, 4096)]:
        small = S.bsp_inputs(b, m, T, 64, seed=1)
        d = {k: L.to_device(np.tile(small[k], (N // 64,) + (1,) * (small[k].ndim - 1)), 1) for k in ARGS}
        lam = torch.full((N,), 0.5, dtype=torch.float64, device="cuda")
        run = lambda: backward.bsp_backward_pass(*[d[k] for k in ARGS], lam, 1, b, m, T, N, 1)
        o = run()
        torch.cuda.synchronize()
        assert int((o["diverge"] != -1).sum()) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        bb = b * b
        rd = bb + b + m * m + m + b * m + bb + bb + b * m + 2 * bb * b + 2 * bb * bb + 2 * bb * m   # doubles read per step
        wr = bb + b + bb + m * b + m                                                              # S, s, tau, K, kff
        gb = 8.0 * (rd + wr) * T * N / 1e9
        out.append(dict(b=b, m=m, T=T, N=N, ms=ms, note="includes output allocation + memset",
                        kernel="lane = problem (bsp_tpp.hip)" if (b in (2, 4) and m in (1, 2)) else "wave = problem, run-time dimensions (gen_wave.hip)",
                        problem_steps_per_s=T * N / ms * 1e3, alg_bytes_per_step=8 * (rd + wr), alg_GB=gb,
                        GBps=gb / ms * 1e3))
        print(json.dumps(out[-1]), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
