"""GPS widening row: the native module `trajopt.gps.core` on the GPU."""
from .core import (augment_cost, backward_pass, backward_pass_status, forward_pass, kl_divergence,  # noqa: F401
                   quad_expectation)
