"""trajopt.bspilqr.core (/root/reference/trajopt/bspilqr/src/util.cpp:206-209): backward_pass, same 19 arguments and 7-tuple."""
from trajopt_amd.bspilqr.core import backward_pass, backward_pass_flag  # noqa: F401
