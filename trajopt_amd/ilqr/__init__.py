"""Mirror of the reference package `trajopt.ilqr` (/root/reference/trajopt/ilqr/__init__.py exports iLQR)."""
from .ilqr import BatchedILQR, iLQR  # noqa: F401
from .streamed import StreamedILQR, auto_sub_batches  # noqa: F401
