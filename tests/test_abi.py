"""The C-ABI library loads without a GPU, exports every symbol include/trajopt_amd.h declares, its structs
have the layout the ctypes mirror assumes, argument validation works, and nothing computes on the CPU."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "trajopt_amd.h")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(trajopt_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree_on_the_symbol_list():
    from trajopt_amd import _lib
    assert declared_functions() == sorted(_lib.EXPORTS)


def test_library_loads_and_exports_every_declared_symbol():
    from trajopt_amd import _lib
    L = _lib.lib()
    for name in declared_functions():
        assert hasattr(L, name), name
    assert L.trajopt_abi_version() == _lib.ABI_VERSION == 5
    assert L.trajopt_error_string(-4).decode().startswith("no HIP device")


def test_struct_layouts_match_the_c_header():
    """Compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    from trajopt_amd import _lib
    from oracle import oracle as O
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "trajopt_amd.h"
int main(void) {
  printf("env %zu %zu %zu %zu\n", sizeof(trajopt_env_t), offsetof(trajopt_env_t, dt), offsetof(trajopt_env_t, lin_A), offsetof(trajopt_env_t, phys));
  printf("io %zu %zu %zu %zu\n", sizeof(trajopt_backward_io_t), offsetof(trajopt_backward_io_t, dV), offsetof(trajopt_backward_io_t, active), offsetof(trajopt_backward_io_t, out_bank));
  printf("opt %zu %zu %zu %zu\n", sizeof(trajopt_ilqr_options_t), offsetof(trajopt_ilqr_options_t, alphas), offsetof(trajopt_ilqr_options_t, keep_value), offsetof(trajopt_ilqr_options_t, no_fusion));
  printf("plan %zu\n", sizeof(trajopt_ilqr_plan_t));
  printf("ws %zu %zu %zu %zu\n", sizeof(trajopt_ilqr_workspace_t), offsetof(trajopt_ilqr_workspace_t, trace_cap), offsetof(trajopt_ilqr_workspace_t, A), offsetof(trajopt_ilqr_workspace_t, out_bank));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = subprocess.check_output([exe]).decode().split("\n")
    E, IO, OP, WS = _lib.EnvStruct, _lib.BackwardIO, _lib.ILQROptions, _lib.ILQRWorkspace
    assert got[0] == "env %d %d %d %d" % (C.sizeof(E), E.dt.offset, E.lin_A.offset, E.phys.offset)
    assert got[1] == "io %d %d %d %d" % (C.sizeof(IO), IO.dV.offset, IO.active.offset, IO.out_bank.offset)
    assert got[2] == "opt %d %d %d %d" % (C.sizeof(OP), OP.alphas.offset, OP.keep_value.offset, OP.no_fusion.offset)
    assert got[3] == "plan %d" % C.sizeof(_lib.ILQRPlan)
    assert got[4] == "ws %d %d %d %d" % (C.sizeof(WS), WS.trace_cap.offset, WS.A.offset, WS.out_bank.offset)
    assert C.sizeof(O.EnvStruct) == C.sizeof(E)   # the oracle's mirror of the same POD


def test_bad_arguments_are_rejected_before_any_device_work():
    from trajopt_amd import _lib
    L = _lib.lib()
    io = _lib.BackwardIO()
    assert L.trajopt_ilqr_backward_pass_batched(C.byref(io), 1, 4, 1, 10, C.c_int64(8), 1, 0, None, None) == -1
    assert L.trajopt_ilqr_backward_pass_batched(None, 1, 4, 1, 10, C.c_int64(8), 1, 0, None, None) == -1
    assert L.trajopt_convert_layout(None, None, C.c_int64(4), C.c_int64(4), 0, 1, None) == -1
    assert L.trajopt_quadratize_batched(None, 10, C.c_int64(1), 1, 0, *([None] * 13)) == -1
    assert L.trajopt_profile_end(None, None) == -1


def test_no_cpu_fallback_exists():
    """Without a GPU the product path must fail loudly (TrajoptError), never compute on the host."""
    import numpy as np
    from trajopt_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible; the fail-loudly path is covered on CPU-only hosts")
    from trajopt_amd import envs
    from trajopt_amd.ilqr import BatchedILQR, core
    assert _lib.device_count() == -4
    with pytest.raises(_lib.TrajoptError):
        core.backward_pass(np.zeros((2, 2, 3)), np.zeros((2, 3)), np.zeros((1, 1, 3)), np.zeros((1, 3)),
                           np.zeros((2, 1, 3)), np.zeros((2, 2, 2)), np.zeros((2, 1, 2)), 1.0, 1, 2, 1, 2)
    with pytest.raises(_lib.TrajoptError):
        BatchedILQR(envs.make("pendulum"), 10, np.zeros((4, 2)))
    with pytest.raises(_lib.TrajoptError):
        envs.make("pendulum").dynamics(np.zeros(2), np.zeros(1))


def test_library_exports_nothing_beyond_the_header():
    """No debug knobs or other undeclared trajopt_* entry points in the product library."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "trajopt_amd", "libtrajopt_amd.so")]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("trajopt_"))
    assert exported == declared_functions()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "trajopt_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
                assert "/root/reference" not in txt or f.endswith((".py", ".hip", ".h"))  # citations only
    out = subprocess.check_output([sys.executable, "-c",
                                   "import sys; sys.path.insert(0, %r); import trajopt_amd, trajopt_amd.ilqr, "
                                   "trajopt_amd.riccati, trajopt_amd.envs, trajopt_amd.sharding; "
                                   "print(any(m.startswith('oracle') for m in sys.modules))" % ROOT])
    assert out.decode().strip() == "False"
