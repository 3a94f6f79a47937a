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
                               (4, 2, 500, 4096, False), (2, 1, 100, 65536, False),
                               # shapes without a lane-per-problem instantiation: the wave-per-problem kernel with run-time
                               # dimensions (gen_wave.hip) - quad pendulum (8, 4) and a 12 x 4 model
                               (8, 4, 100, 4096, False), (8, 4, 100, 16384, False), (12, 4, 100, 4096, False)]:
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
                        kernel="lane = problem (gps_tpp.hip)" if (n, m) in ((2, 1), (3, 1), (4, 1), (6, 1), (2, 2), (4, 2)) else "wave = problem, run-time dimensions (gen_wave.hip)",
                        problem_steps_per_s=T * N / ms * 1e3, alg_GB=gb, GBps=gb / ms * 1e3))
        print(json.dumps(out[-1]), flush=True)
    out.append(dual_chain(4, 1, 100, 16384))
    print(json.dumps(out[-1]), flush=True)
    json.dump(out, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None


def dual_chain(n, m, T, N):
    """One MBGPS dual evaluation (gps/mbgps.py:173-195) for N problems, arrays resident: augment_cost -> backward_pass
    -> forward_pass -> quad_expectation -> kl_divergence; per-phase device time with events on the launch stream."""
    small = S.gps_inputs(n, m, T, 64, seed=2)
    rep = lambda a: L.to_device(np.tile(a, (N // 64,) + (1,) * (a.ndim - 1)), 1)
    d = {k: rep(small[k]) for k in ARGS}
    o = backward.gps_backward_pass(*[d[k] for k in ARGS], n, m, T, N, 1, full=False)   # a "last" controller
    ctl = (o["K"], o["kff"], o["sigma_ctl"])
    rng = np.random.default_rng(0)
    mu0 = L.to_device(rng.standard_normal((N, n)), 1)
    s0 = L.to_device(np.tile(1e-2 * np.eye(n), (N, 1, 1)), 1)
    names = ("augment_cost", "backward_pass", "forward_pass", "quad_expectation", "kl_divergence")

    def run(ev=None):
        mark = (lambda i: ev[i].record()) if ev else (lambda i: None)
        mark(0)
        ag = backward.gps_augment_cost(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["c0"], *ctl, d["alpha"], n, m, T, N)
        mark(1)
        bw = backward.gps_backward_pass(ag["Cxx"], ag["cx"], ag["Cuu"], ag["cu"], ag["Cxu"], ag["c0"], d["A"], d["B"],
                                        d["c"], d["sigma_dyn"], d["alpha"], n, m, T, N, 1, full=False)
        mark(2)
        fw = backward.gps_forward_pass(mu0, s0, d["A"], d["B"], d["c"], d["sigma_dyn"], bw["K"], bw["kff"],
                                       bw["sigma_ctl"], n, m, T, N)
        mark(3)
        dual = backward.gps_quad_expectation(mu0, s0, bw["V"][0], bw["v"][0], bw["v0"][0], n, N)
        mark(4)
        kl, st = backward.gps_kl_divergence(bw["K"], bw["kff"], bw["sigma_ctl"], *ctl, fw["mu_x"], fw["sigma_x"], n, m, T, N)
        mark(5)
        return dual, kl, bw["diverge"], st

    dual, kl, div, st = run()
    torch.cuda.synchronize()
    assert int((div != -1).sum()) == 0 and int(st.abs().sum()) == 0 and bool(torch.isfinite(dual).all())
    reps, acc = 10, [0.0] * 5
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        run(ev)
        torch.cuda.synchronize()
        for i in range(5):
            acc[i] += ev[i].elapsed_time(ev[i + 1]) / reps
    total = sum(acc)
    return dict(what="MBGPS dual evaluation chain", n=n, m=m, T=T, N=N, ms=total, phase_ms=dict(zip(names, acc)),
                note="includes output allocation", problem_steps_per_s=T * N / total * 1e3)


if __name__ == "__main__":
    main()
