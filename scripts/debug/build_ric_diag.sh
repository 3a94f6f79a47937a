#!/bin/bash
# Builds scripts/debug/libdiag.so: the library with ric_mfma64.hip compiled with -DRIC_DIAG=1 (per-phase clock ticks
# instead of results; developer tool, see ric_phases.py).  Extra flags for that one file: "$@".
set -e
cd "$(dirname "$0")/../../trajopt_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DRIC_DIAG=1 "$@" -c ric_mfma64.hip -o /tmp/ric_diag.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/debug/libdiag.so $(ls *.o | grep -v ric_mfma64.o) /tmp/ric_diag.o
