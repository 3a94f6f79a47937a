"""Time the max-entropy backward pass on resident inputs: ms per launch and algorithmic GB/s."""
import json
import sys

import numpy as np
import torch

from trajopt_amd import backward, layout as L, synthetic as S

ARGS = ("Cxx", "cx", "Cuu", "cu", "Cxu", "c0", "A", "B", "c", "sigma_dyn", "alpha")


def main():
    out = []
    for (n, m, T, N, full) in [(4, 1, 500, 4096, False), (4, 1, 500, 4096, True), (6, 1, 500, 4096, False),
                               (4, 2, 500, 4096, False), (2, 1, 100, 65536, False)]:
        small = S.gps_inputs(n, m, T, 64, seed=1)
        d = {k: L.to_device(np.tile(small[k], (N // 64,) + (1,) * (small[k].ndim - 1)), 1) for k in ARGS}
        run = lambda: backward.gps_backward_pass(*[d[k] for k in ARGS], n, m, T, N, 1, full=full)
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
        rd = 2 * n * n + n * m + n + n * n + m * m + n * m + n + m + 2          # per step, doubles read
        wr = m * n + m + m * m + n * n + n + 1 + (n * n + m * n + m * m + n + m + 1 if full else 0)
        gb = 8.0 * (rd + wr) * T * N / 1e9
        out.append(dict(n=n, m=m, T=T, N=N, full=full, ms=ms, note="includes output allocation + memset",
                        problem_steps_per_s=T * N / ms * 1e3, alg_GB=gb, GBps=gb / ms * 1e3))
        print(json.dumps(out[-1]), flush=True)
    json.dump(out, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None


if __name__ == "__main__":
    main()
