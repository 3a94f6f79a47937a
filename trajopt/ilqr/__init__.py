"""trajopt.ilqr (/root/reference/trajopt/ilqr/__init__.py: `from .ilqr import iLQR`)."""
from trajopt_amd.ilqr import BatchedILQR, StreamedILQR, iLQR  # noqa: F401
from . import core, objects  # noqa: F401
