"""trajopt.gps - only the native module `core` of the reference's package is rebuilt; the MBGPS / MFGPS
drivers (/root/reference/trajopt/gps/__init__.py) are out of scope."""
from . import core  # noqa: F401
