"""Batch layouts of include/trajopt_amd.h, expressed as torch / NumPy shapes.

A per-problem array of the reference has shape (rows, cols, T) (or (rows, T)) and is Fortran-ordered
(/root/reference/trajopt/ilqr/src/util.cpp:10,21).  A batch of N problems is handled on the host side
as a NumPy array of shape (N, rows, cols, T) / (N, rows, T) ("logical" shape, any strides) and on the
device as a C-contiguous torch tensor of shape

    PROBLEM_MAJOR : (N, T, cols, rows)   [p slowest]   == N reference-layout problems back to back
    PROBLEM_MINOR : (T, cols, rows, N)   [p fastest]   == struct-of-arrays, coalesced across problems
"""
import numpy as np

from ._lib import LAYOUT_PROBLEM_MAJOR, LAYOUT_PROBLEM_MINOR


def device_shape(logical_shape, layout):
    """logical (N, rows[, cols], T) -> C-contiguous device shape."""
    N, rest = logical_shape[0], tuple(logical_shape[1:])
    rev = tuple(reversed(rest))
    return (N,) + rev if layout == LAYOUT_PROBLEM_MAJOR else rev + (N,)


def to_device(a, layout, device="cuda"):
    """NumPy logical array (N, rows[, cols], T) -> torch float64 tensor in `layout`."""
    import torch
    a = np.asarray(a, dtype=np.float64)
    nd = a.ndim
    if layout == LAYOUT_PROBLEM_MAJOR:
        perm = (0,) + tuple(range(nd - 1, 0, -1))
    else:
        perm = tuple(range(nd - 1, -1, -1))
    return torch.from_numpy(np.ascontiguousarray(a.transpose(perm))).to(device)


def from_device(t, layout):
    """Inverse of to_device: torch tensor in `layout` -> NumPy logical array (N, rows[, cols], T)."""
    a = t.detach().cpu().numpy()
    nd = a.ndim
    if layout == LAYOUT_PROBLEM_MAJOR:
        perm = (0,) + tuple(range(nd - 1, 0, -1))
    else:
        perm = tuple(range(nd - 1, -1, -1))
    return a.transpose(perm)


def empty(logical_shape, layout, device="cuda", zero=False):
    import torch
    shp = device_shape(logical_shape, layout)
    return (torch.zeros if zero else torch.empty)(shp, dtype=torch.float64, device=device)
