"""Mirror of the reference package `trajopt.riccati`."""
from .riccati import BatchedRiccati, Riccati  # noqa: F401
