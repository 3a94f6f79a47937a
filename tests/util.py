"""Shared helpers of the parity tests."""
import numpy as np

TOL = 1e-9  # north_star: gains, value function, trajectory cost within 1e-9 relative (fp64)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    den = max(np.max(np.abs(b)), 1e-300)
    d = np.abs(a - b)
    if np.isnan(d).any():
        return np.inf
    return float(np.max(d) / den)


def assert_close(a, b, name="", tol=TOL):
    e = rel_err(a, b)
    assert e <= tol, "%s: relative error %.3e > %.1e" % (name, e, tol)
