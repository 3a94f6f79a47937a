"""trajopt.bspilqr - only the native module `core` is rebuilt."""
from . import core  # noqa: F401
