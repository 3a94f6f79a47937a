"""Per-phase clock ticks of the C5 Riccati sweep (developer tool).  Needs a library built with -DRIC_DIAG=1:
    cd trajopt_amd/csrc && hipcc ... -DRIC_DIAG=1 ... -o ../../scripts/debug/libdiag.so ; TRAJOPT_AMD_SO=scripts/debug/libdiag.so python scripts/debug/ric_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajopt_amd import backward
from bench_backward import make
n, m, T, N = 64, 16, 1000, 1024
d = make(n, m, T, N, 0, True)
o = backward.riccati_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], d["c"], n, m, T, N, 0)
torch.cuda.synchronize()
v = o["v"].cpu().numpy().reshape(N, -1)[:, :7]   # PROBLEM_MAJOR: [p][t][e], ticks in slice 0
tot = v.sum(axis=1).mean()
names = ["Ph1 P=V'F", "barrier 1", "Ph2 Q=C+F'P", "barrier 2", "Ph3 LDL', solves", "barrier 3", "Ph4 V, outputs"]
for k, nm in enumerate(names):
    print("%-18s %8.1f ticks/step  %5.1f %%" % (nm, v[:, k].mean() / T, 100 * v[:, k].mean() / tot))
print("total %.1f ticks/step" % (tot / T))
