"""trajopt.riccati (/root/reference/trajopt/riccati/__init__.py: `from .riccati import Riccati`)."""
from trajopt_amd.riccati import BatchedRiccati, Riccati  # noqa: F401
from . import objects  # noqa: F401
