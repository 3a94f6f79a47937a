"""trajopt.gps.core (/root/reference/trajopt/gps/src/util.cpp:376-383): kl_divergence, quad_expectation, augment_cost,
forward_pass, backward_pass with the reference's argument lists and tuples."""
from trajopt_amd.gps.core import (augment_cost, backward_pass, backward_pass_status, forward_pass,  # noqa: F401
                                  kl_divergence, quad_expectation)
