"""GPS widening row: the max-entropy backward pass of trajopt.gps.core on the GPU."""
from .core import backward_pass, backward_pass_status  # noqa: F401
