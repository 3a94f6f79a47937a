"""Belief-space iLQR widening row: the native backward pass of trajopt.bspilqr.core on the GPU."""
from .core import backward_pass, backward_pass_flag  # noqa: F401
