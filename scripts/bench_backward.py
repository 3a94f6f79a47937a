"""Microbenchmark of the batched backward passes alone (developer tool; bench.py is the contract).
  --variant ilqr    : util.cpp recursion (rows A1), full and gains-only outputs
  --variant riccati : riccati.py sweep with affine c (row A9; BASELINE config C5 at --n 64 --m 16 --T 1000 --N 1024)
Inputs are generated on the device (SPD Cxx / Cuu by construction)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from trajopt_amd import backward  # noqa: E402
from trajopt_amd.synthetic import device_backward_inputs  # noqa: E402


def make(n, m, T, N, layout, affine, dev="cuda"):
    return device_backward_inputs(n, m, T, N, layout, affine=affine, dev=dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="ilqr")
    ap.add_argument("--n", type=int, default=4); ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--T", type=int, default=500); ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--layout", type=int, default=1); ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--kernel", type=int, default=0, help="trajopt_backward_io_t::kernel (0 = the library's choice)")
    a = ap.parse_args()
    n, m, T, N = a.n, a.m, a.T, a.N
    d = make(n, m, T, N, a.layout, a.variant == "riccati")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if a.variant == "riccati":
        call = lambda out=None: backward.riccati_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"],
                                                               d["c"], n, m, T, N, a.layout, out=out)
        o = call()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            o = call(o)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        b = 8 * (3 * n * n + 3 * n * m + m * m + 4 * n + 2 * m)   # in A,B,c,Cxx,Cuu,Cxu,cx,cu; out K,kff,V,v
        fl = 4 * n ** 3 + 10 * n * n * m + 6 * n * m * m + 2 * m ** 3 + 2 * n * n
        print("riccati n=%d m=%d T=%d N=%d layout=%d: %.3f ms  %.1f GB/s  %.2f TFLOP/s  bad=%d finite=%s" % (
            n, m, T, N, a.layout, ms, N * T * b / ms / 1e6, N * T * fl / ms / 1e9, int((o["status"] != 0).sum()),
            bool(torch.isfinite(o["K"]).all())))
        return
    lam = torch.ones(N, dtype=torch.float64, device="cuda")
    for full in (True, False):
        out = {}
        for _ in range(2):
            backward.ilqr_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], lam, 1, n, m, T,
                                        N, a.layout, full=full, out=out, kernel=a.kernel)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            backward.ilqr_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], lam, 1, n, m, T,
                                        N, a.layout, full=full, out=out, kernel=a.kernel)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        b = 8 * (4 * n * n + 4 * n * m + 2 * m * m + 3 * n + 3 * m) if full else \
            8 * (2 * n * n + 3 * n * m + m * m + n + 2 * m)
        print("ilqr n=%d m=%d T=%d N=%d layout=%d full=%d: %.3f ms  %.1f GB/s  (%.2f G problem-steps/s) diverged=%d" % (
            n, m, T, N, a.layout, full, ms, N * T * b / ms / 1e6, N * T / ms / 1e6, int((out["diverge"] >= 0).sum())))


if __name__ == "__main__":
    main()
