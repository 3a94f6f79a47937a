"""trajopt.ilqr.core - the module /root/reference/trajopt/ilqr/src/util.cpp:185-188 builds with pybind11; here
`backward_pass` (same 12 positional arguments, same 11-tuple) runs the HIP kernels through trajopt_ilqr_backward_pass_host."""
from trajopt_amd.ilqr.core import backward_pass, backward_pass_flag  # noqa: F401
