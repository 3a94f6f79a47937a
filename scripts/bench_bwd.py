"""Microbenchmark of the batched backward pass alone (developer tool; bench.py is the contract)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajopt_amd import backward
from trajopt_amd.layout import device_shape


def make(n, m, T, N, layout, dev="cuda"):
    g = torch.Generator(device=dev); g.manual_seed(1337)
    def rn(*logical):
        return torch.randn(device_shape(logical, layout), generator=g, dtype=torch.float64, device=dev)
    eye = lambda k, Tn: (torch.eye(k, dtype=torch.float64, device=dev).reshape((1, k, k, 1)) if layout == 0 else
                         torch.eye(k, dtype=torch.float64, device=dev).reshape((1, k, k, 1)))
    def with_eye(x, k, scale):
        # add scale*I along the (cols, rows) axes of the device shape
        I = torch.eye(k, dtype=torch.float64, device=dev)
        if layout == 0:
            return x + scale * I.reshape(1, 1, k, k)
        return x + scale * I.reshape(1, k, k, 1)
    A = with_eye(0.01 * rn(N, n, n, T) / n ** 0.5, n, 1.0)
    B = 0.01 * rn(N, n, m, T)
    Cxx = with_eye(0.0 * rn(N, n, n, T + 1), n, 1.0)   # SPD, cheap to build at scale
    Cuu = with_eye(0.0 * rn(N, m, m, T + 1), m, 1e-2)
    Cxu = 0.01 * rn(N, n, m, T + 1)
    cx, cu = rn(N, n, T + 1), rn(N, m, T + 1)
    return dict(A=A, B=B, Cxx=Cxx, Cuu=Cuu, Cxu=Cxu, cx=cx, cu=cu)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4); ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--T", type=int, default=500); ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--same-step", type=int, default=0); ap.add_argument("--layout", type=int, default=1); ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    n, m, T, N = a.n, a.m, a.T, a.N
    from trajopt_amd._lib import lib
    lib().trajopt_debug_tpp_same_step(a.same_step)
    d = make(n, m, T, N, a.layout)
    lam = torch.ones(N, dtype=torch.float64, device="cuda")
    for full in (True, False):
        out = {}
        for _ in range(3):
            backward.ilqr_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], lam, 1, n, m, T, N,
                                        a.layout, full=full, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            backward.ilqr_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], lam, 1, n, m, T, N,
                                        a.layout, full=full, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        bytes_full = 8 * (4 * n * n + 4 * n * m + 2 * m * m + 3 * n + 3 * m)
        bytes_gain = 8 * (2 * n * n + 3 * n * m + m * m + n + 2 * m)
        b = bytes_full if full else bytes_gain
        print("n=%d m=%d T=%d N=%d layout=%d full=%d: %.3f ms  %.1f GB/s  (%.2f G problem-steps/s) diverged=%d" % (
            n, m, T, N, a.layout, full, ms, N * T * b / ms / 1e6, N * T / ms / 1e6, int((out["diverge"] >= 0).sum())))


if __name__ == "__main__":
    main()
