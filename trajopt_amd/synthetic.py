"""Synthetic inputs for the backward pass in isolation.

Per problem p and step t:  A = I + dt*G/sqrt(n), G ~ N(0,1), dt = 0.01;  B = dt*N(0,1);
Cxx = M^T M / n + 1e-3 I;  Cuu = S^T S / m + 1e-2 I;  Cxu = 1e-2 N(0,1);  cx, cu ~ N(0,1);
c ~ 1e-2 N(0,1) (Riccati form only).  Seed: numpy Generator(PCG64(1337)), the seed the reference
examples use (/root/reference/examples/riccati/lqr.py:6).

Returned arrays are "problem-major": shape (N, rows, cols, T) / (N, rows, T) whose per-problem slices
are Fortran-ordered exactly like the reference's arguments (util.cpp:10,21); `to_layout` converts.
"""
import numpy as np

SEED = 1337


def backward_inputs(n, m, T, N, seed=SEED, affine=False, dt=0.01):
    rng = np.random.Generator(np.random.PCG64(seed))

    def fo(shape):  # (N, T, cols, rows) C-order  ==  per-problem F-order (rows, cols, T)
        return rng.standard_normal(shape)

    G = fo((N, T, n, n))
    A = np.eye(n)[None, None] + dt * G / np.sqrt(n)
    B = dt * fo((N, T, m, n))
    M = fo((N, T + 1, n, n))
    Cxx = np.einsum("ptki,ptkj->ptij", M, M) / n + 1e-3 * np.eye(n)[None, None]
    S = fo((N, T + 1, m, m))
    Cuu = np.einsum("ptki,ptkj->ptij", S, S) / m + 1e-2 * np.eye(m)[None, None]
    Cxu = 1e-2 * fo((N, T + 1, m, n))
    cx = fo((N, T + 1, n))
    cu = fo((N, T + 1, m))
    out = dict(A=A, B=B, Cxx=Cxx, Cuu=Cuu, Cxu=Cxu, cx=cx, cu=cu)
    if affine:
        out["c"] = 1e-2 * fo((N, T, n))
    # stored as (N, T, cols, rows): present them as (N, rows, cols, T) views with the same memory
    res = {}
    for k, a in out.items():
        if a.ndim == 4:
            res[k] = np.ascontiguousarray(a).transpose(0, 3, 2, 1)
        else:
            res[k] = np.ascontiguousarray(a).transpose(0, 2, 1)
    return res


def problem(stack, p):
    """Reference-layout (Fortran-ordered) arrays of problem p."""
    return {k: np.asfortranarray(a[p]) for k, a in stack.items()}


def gps_inputs(n, m, T, N, seed=SEED):
    """Inputs of the max-entropy backward pass (gps/src/util.cpp:272): backward_inputs(affine=True) plus
    c0 ~ N(0,1) (N, T+1), sigma_dyn = 1e-4 (R^T R / n + I) (N, n, n, T), alpha ~ U[0.5, 50] (N, T)."""
    out = backward_inputs(n, m, T, N, seed=seed, affine=True)
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    out["c0"] = rng.standard_normal((N, T + 1))
    R = rng.standard_normal((N, T, n, n))
    S = 1e-4 * (np.einsum("ptki,ptkj->ptij", R, R) / n + np.eye(n)[None, None])
    out["sigma_dyn"] = np.ascontiguousarray(S).transpose(0, 3, 2, 1)
    out["alpha"] = rng.uniform(0.5, 50.0, size=(N, T))
    return out


def bsp_inputs(b, m, T, N, seed=SEED, coupling=1e-2):
    """Inputs of the belief-space backward pass (bspilqr/src/util.cpp:83-91): the iLQR stack renamed
    (Q, q, R, r, P <- Cxx, cx, Cuu, cu, Cxu; F, G <- A, B) plus p ~ N(0,1) (N, b^2, T+1) and the belief-covariance
    couplings T, X (b^2, b), U, Y (b^2, b^2), V, Z (b^2, m) ~ coupling * N(0,1), all with T slices."""
    base = backward_inputs(b, m, T, N, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    bb = b * b
    out = dict(Q=base["Cxx"], q=base["cx"], R=base["Cuu"], r=base["cu"], P=base["Cxu"], F=base["A"], G=base["B"])
    out["p"] = np.ascontiguousarray(rng.standard_normal((N, T + 1, bb))).transpose(0, 2, 1)
    for k, cols in (("T", b), ("U", bb), ("V", m), ("X", b), ("Y", bb), ("Z", m)):
        out[k] = np.ascontiguousarray(coupling * rng.standard_normal((N, T, cols, bb))).transpose(0, 3, 2, 1)
    return out


def device_backward_inputs(n, m, T, N, layout, affine=False, dev="cuda", seed=SEED):
    """Dense backward-pass inputs generated ON THE DEVICE in `layout` (benchmarks at sizes the host generator above
    would take minutes for): A = I + 0.01 G / sqrt(n), B = 0.01 N(0,1), Cxx = I, Cuu = 1e-2 I, Cxu = 0.01 N(0,1),
    cx, cu ~ N(0,1), c = 0.01 N(0,1) (Riccati form) - SPD Cxx / Cuu by construction.  -> dict of torch tensors."""
    import torch
    from .layout import device_shape
    g = torch.Generator(device=dev)
    g.manual_seed(seed)

    def rn(scale, *logical):
        shp = device_shape(logical, layout)
        if scale == 0.0:
            return torch.zeros(shp, dtype=torch.float64, device=dev)
        x = torch.randn(shp, generator=g, dtype=torch.float64, device=dev)
        return x.mul_(scale)

    def add_eye(x, k, scale):
        I = scale * torch.eye(k, dtype=torch.float64, device=dev)
        x += I.reshape(1, 1, k, k) if layout == 0 else I.reshape(1, k, k, 1)
        return x

    d = dict(A=add_eye(rn(0.01 / n ** 0.5, N, n, n, T), n, 1.0), B=rn(0.01, N, n, m, T),
             Cxx=add_eye(rn(0.0, N, n, n, T + 1), n, 1.0), Cuu=add_eye(rn(0.0, N, m, m, T + 1), m, 1e-2),
             Cxu=rn(0.01, N, n, m, T + 1), cx=rn(1.0, N, n, T + 1), cu=rn(1.0, N, m, T + 1))
    if affine:
        d["c"] = rn(0.01, N, n, T)
    return d
