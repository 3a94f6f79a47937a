"""Batched backward passes on device tensors (rows A1 and A9 of the hot path).

Thin plumbing over the C ABI: torch only provides device memory and the current stream.
"""
import ctypes as C

from . import _lib
from ._lib import BackwardIO, BspBackwardIO, GpsBackwardIO, LambdaSchedule, LAYOUT_PROBLEM_MAJOR, LAYOUT_PROBLEM_MINOR, check, dptr, lib, stream_ptr
from .layout import empty


def _chk(t, numel, name):
    import torch
    if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or t.numel() != numel:
        raise ValueError("%s must be a contiguous CUDA float64 tensor with %d elements" % (name, numel))


def ilqr_backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, dm_state, dm_act, nb_steps, nb_problems, layout,
                       full=True, retry=False, sched=None, dlmbda=None, uref=None, active=None, out=None, cost=None,
                       kernel=0):
    """Regularised Riccati recursion of /root/reference/trajopt/ilqr/src/util.cpp:83-182 for a batch.

    Inputs are device tensors in `layout` (see layout.py); `lmbda` is a [N] tensor (updated in place when
    retry=True).  Returns a dict with K, kff, dV, diverge (and Qxx, Qux, Quu, qx, qu, V, v when full=True,
    g_norm when uref is given).  diverge[p] = -1 when every step passed the PD test.
    cost=(w, gxx, guu): structured cost (trajopt_backward_io_t::cost_w) - Cxx[t] = w[t] diag(gxx), Cuu = diag(guu),
    Cxu = 0 for every problem, w a [T+1] device tensor; Cxx, Cuu, Cxu are then ignored and may be None.
    kernel: _lib.KERNEL_* family (0 = the library's choice; the parity tests force each family in turn).
    """
    import torch
    _lib.require_gpu()
    n, m, T, N = dm_state, dm_act, nb_steps, nb_problems
    dense = () if cost is not None else ((Cxx, n * n * (T + 1), "Cxx"), (Cuu, m * m * (T + 1), "Cuu"),
                                         (Cxu, n * m * (T + 1), "Cxu"))
    for t, cnt, nm in dense + ((cx, n * (T + 1), "cx"), (cu, m * (T + 1), "cu"), (A, n * n * T, "A"),
                               (B, n * m * T, "B")):
        _chk(t, cnt * N, nm)
    _chk(lmbda, N, "lmbda")
    dev = A.device
    o = {} if out is None else out

    def need(key, *logical):   # allocated (zero-filled) only when the caller's `out` does not hold it yet
        if key not in o:
            o[key] = empty((N,) + logical, layout, dev, zero=True)

    need("K", m, n, T)
    need("kff", m, T)
    need("dV", 2)
    if "diverge" not in o:
        o["diverge"] = torch.empty(N, dtype=torch.int32, device=dev)
    if full:
        need("Qxx", n, n, T)
        need("Qux", m, n, T)
        need("Quu", m, m, T)
        need("qx", n, T)
        need("qu", m, T)
        need("V", n, n, T + 1)
        need("v", n, T + 1)
    if uref is not None and "g_norm" not in o:
        o["g_norm"] = torch.empty(N, dtype=torch.float64, device=dev)
    io = BackwardIO()
    io.cx, io.cu, io.A, io.B = map(dptr, (cx, cu, A, B))
    if cost is None:
        io.Cxx, io.Cuu, io.Cxu = map(dptr, (Cxx, Cuu, Cxu))
    else:
        w, gxx, guu = cost
        if n > _lib.MAX_STATE or m > _lib.MAX_ACT:
            raise _lib.TrajoptError("the structured cost holds at most %d state and %d action weights" % (_lib.MAX_STATE, _lib.MAX_ACT))
        if w.dtype != torch.float64 or not w.is_cuda or not w.is_contiguous() or w.numel() != T + 1:
            raise ValueError("cost weighting must be a contiguous CUDA float64 tensor with %d elements" % (T + 1))
        io.cost_w = dptr(w)
        for i in range(n):
            io.cost_gxx[i] = float(gxx[i])
        for j in range(m):
            io.cost_guu[j] = float(guu[j])
    io.lmbda, io.dlmbda = dptr(lmbda), dptr(dlmbda)
    for k in ("Qxx", "Qux", "Quu", "qx", "qu", "V", "v"):
        setattr(io, k, dptr(o.get(k)) if full else None)
    io.dV, io.K, io.kff, io.diverge = dptr(o["dV"]), dptr(o["K"]), dptr(o["kff"]), dptr(o["diverge"])
    io.uref, io.g_norm, io.active = dptr(uref), dptr(o.get("g_norm")), dptr(active)
    io.kernel = int(kernel)
    s = sched if sched is not None else LambdaSchedule(1e-6, 1e6, 1.6)
    check(lib().trajopt_ilqr_backward_pass_batched(C.byref(io), int(reg), n, m, T, C.c_int64(N), int(layout),
                                                   int(bool(retry)), C.byref(s), stream_ptr()))
    return o


def riccati_backward_pass(Cxx, cx, Cuu, cu, Cxu, A, B, c, dm_state, dm_act, nb_steps, nb_problems, layout,
                          with_status=True, out=None):
    """TV-LQR sweep of /root/reference/trajopt/riccati/riccati.py:100-126 for a batch -> K, kff, V, v, status.
    with_status=False passes status = NULL (then at most one such call per GPU at a time: include/trajopt_amd.h).
    out: a dict returned by an earlier call of the same shape - its tensors are written again instead of allocating
    (and zero-filling) new ones (C5: 42 GB)."""
    import torch
    _lib.require_gpu()
    n, m, T, N = dm_state, dm_act, nb_steps, nb_problems
    for t, cnt, nm in ((Cxx, n * n * (T + 1), "Cxx"), (cx, n * (T + 1), "cx"), (Cuu, m * m * (T + 1), "Cuu"),
                       (cu, m * (T + 1), "cu"), (Cxu, n * m * (T + 1), "Cxu"), (A, n * n * T, "A"),
                       (B, n * m * T, "B"), (c, n * T, "c")):
        _chk(t, cnt * N, nm)
    dev = Cxx.device
    o = out if out is not None else dict(
        K=empty((N, m, n, T), layout, dev, zero=True), kff=empty((N, m, T), layout, dev, zero=True),
        V=empty((N, n, n, T + 1), layout, dev, zero=True), v=empty((N, n, T + 1), layout, dev, zero=True),
        status=torch.zeros(N, dtype=torch.int32, device=dev) if with_status else None)
    check(lib().trajopt_riccati_backward_pass_batched(
        dptr(Cxx), dptr(cx), dptr(Cuu), dptr(cu), dptr(Cxu), dptr(A), dptr(B), dptr(c), n, m, T, C.c_int64(N),
        int(layout), dptr(o["V"]), dptr(o["v"]), dptr(o["K"]), dptr(o["kff"]), dptr(o["status"]), stream_ptr()))
    return o


def gps_backward_pass(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha, dm_state, dm_act, nb_steps, nb_problems,
                      layout=LAYOUT_PROBLEM_MINOR, full=True):
    """Max-entropy backward pass of /root/reference/trajopt/gps/src/util.cpp:272-374 for a batch.

    Device tensors in PROBLEM_MINOR; alpha is (N, T) in that layout (one temperature per step and problem, as
    mbgps.py's pass_alpha_as_vector hands over).  Returns K, kff, sigma_ctl, V, v, v0, diverge (-1 = completed),
    status, and Qxx ... q0 when full=True.
    """
    import torch
    _lib.require_gpu()
    n, m, T, N = dm_state, dm_act, nb_steps, nb_problems
    for t, cnt, nm in ((Cxx, n * n * (T + 1), "Cxx"), (cx, n * (T + 1), "cx"), (Cuu, m * m * (T + 1), "Cuu"),
                       (cu, m * (T + 1), "cu"), (Cxu, n * m * (T + 1), "Cxu"), (c0, T + 1, "c0"), (A, n * n * T, "A"),
                       (B, n * m * T, "B"), (c, n * T, "c"), (sigma_dyn, n * n * T, "sigma_dyn"), (alpha, T, "alpha")):
        _chk(t, cnt * N, nm)
    dev = A.device
    o = dict(K=empty((N, m, n, T), layout, dev, zero=True), kff=empty((N, m, T), layout, dev, zero=True),
             sigma_ctl=empty((N, m, m, T), layout, dev, zero=True), V=empty((N, n, n, T + 1), layout, dev, zero=True),
             v=empty((N, n, T + 1), layout, dev, zero=True), v0=empty((N, T + 1), layout, dev, zero=True),
             diverge=torch.empty(N, dtype=torch.int32, device=dev), status=torch.empty(N, dtype=torch.int32, device=dev))
    if full:
        o.update(Qxx=empty((N, n, n, T), layout, dev, zero=True), Qux=empty((N, m, n, T), layout, dev, zero=True),
                 Quu=empty((N, m, m, T), layout, dev, zero=True), qx=empty((N, n, T), layout, dev, zero=True),
                 qu=empty((N, m, T), layout, dev, zero=True), q0=empty((N, T), layout, dev, zero=True))
    io = GpsBackwardIO()
    for k, t in (("Cxx", Cxx), ("cx", cx), ("Cuu", Cuu), ("cu", cu), ("Cxu", Cxu), ("c0", c0), ("A", A), ("B", B),
                 ("c", c), ("sigma_dyn", sigma_dyn), ("alpha", alpha)):
        setattr(io, k, dptr(t))
    for k in ("Qxx", "Qux", "Quu", "qx", "qu", "q0", "V", "v", "v0", "K", "kff", "sigma_ctl"):
        setattr(io, k, dptr(o.get(k)))
    io.diverge, io.status = dptr(o["diverge"]), dptr(o["status"])
    check(lib().trajopt_gps_backward_pass_batched(C.byref(io), n, m, T, C.c_int64(N), int(layout), stream_ptr()))
    return o


def _gps_check(tensors, N):
    for t, cnt, nm in tensors:
        _chk(t, cnt * N, nm)


def gps_forward_pass(mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, dm_state, dm_act, nb_steps, nb_problems,
                     layout=LAYOUT_PROBLEM_MINOR):
    """Gaussian moment propagation of gps/src/util.cpp:195-269 for a batch -> dict mu_x, sigma_x, mu_u, sigma_u,
    mu_xu, sigma_xu.  mu_x0 / sigma_x0 are (N, n) / (N, n, n) in `layout` (device shapes (n, N) / (n, n, N))."""
    _lib.require_gpu()
    n, m, T, N = dm_state, dm_act, nb_steps, nb_problems
    _gps_check(((mu_x0, n, "mu_x0"), (sigma_x0, n * n, "sigma_x0"), (A, n * n * T, "A"), (B, n * m * T, "B"),
                (c, n * T, "c"), (sigma_dyn, n * n * T, "sigma_dyn"), (K, n * m * T, "K"), (kff, m * T, "kff"),
                (sigma_ctl, m * m * T, "sigma_ctl")), N)
    dev, w = A.device, n + m
    o = dict(mu_x=empty((N, n, T + 1), layout, dev), sigma_x=empty((N, n, n, T + 1), layout, dev),
             mu_u=empty((N, m, T), layout, dev), sigma_u=empty((N, m, m, T), layout, dev),
             mu_xu=empty((N, w, T + 1), layout, dev), sigma_xu=empty((N, w, w, T + 1), layout, dev))
    check(lib().trajopt_gps_forward_pass_batched(
        *map(dptr, (mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl)), n, m, T, C.c_int64(N), int(layout),
        *[dptr(o[k]) for k in ("mu_x", "sigma_x", "mu_u", "sigma_u", "mu_xu", "sigma_xu")], stream_ptr()))
    return o


def gps_augment_cost(Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, dm_state, dm_act, nb_steps, nb_problems,
                     layout=LAYOUT_PROBLEM_MINOR):
    """gps/src/util.cpp:136-193 for a batch -> dict Cxx, cx, Cuu, cu, Cxu, c0 (T+1 slices) and status."""
    import torch
    _lib.require_gpu()
    n, m, T, N = dm_state, dm_act, nb_steps, nb_problems
    _gps_check(((Cxx, n * n * (T + 1), "Cxx"), (cx, n * (T + 1), "cx"), (Cuu, m * m * (T + 1), "Cuu"),
                (cu, m * (T + 1), "cu"), (Cxu, n * m * (T + 1), "Cxu"), (c0, T + 1, "c0"), (K, n * m * T, "K"),
                (kff, m * T, "kff"), (sigma_ctl, m * m * T, "sigma_ctl"), (alpha, T, "alpha")), N)
    dev = Cxx.device
    o = dict(Cxx=empty((N, n, n, T + 1), layout, dev), cx=empty((N, n, T + 1), layout, dev),
             Cuu=empty((N, m, m, T + 1), layout, dev), cu=empty((N, m, T + 1), layout, dev),
             Cxu=empty((N, n, m, T + 1), layout, dev), c0=empty((N, T + 1), layout, dev),
             status=torch.empty(N, dtype=torch.int32, device=dev))
    check(lib().trajopt_gps_augment_cost_batched(
        *map(dptr, (Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha)), n, m, T, C.c_int64(N), int(layout),
        *[dptr(o[k]) for k in ("Cxx", "cx", "Cuu", "cu", "Cxu", "c0", "status")], stream_ptr()))
    return o


def gps_kl_divergence(p_K, p_kff, p_sigma_ctl, q_K, q_kff, q_sigma_ctl, mu_x, sigma_x, dm_state, dm_act, nb_steps,
                      nb_problems, layout=LAYOUT_PROBLEM_MINOR):
    """gps/src/util.cpp:83-121 for a batch -> (kl (N, T) in `layout`, status).  mu_x / sigma_x may hold T or T+1
    slices (time is the slowest device axis in PROBLEM_MINOR, so the first T are a prefix)."""
    import torch
    _lib.require_gpu()
    n, m, T, N = dm_state, dm_act, nb_steps, nb_problems
    _gps_check(((p_K, n * m * T, "p_K"), (p_kff, m * T, "p_kff"), (p_sigma_ctl, m * m * T, "p_sigma_ctl"),
                (q_K, n * m * T, "q_K"), (q_kff, m * T, "q_kff"), (q_sigma_ctl, m * m * T, "q_sigma_ctl")), N)
    for t, per, nm in ((mu_x, n, "mu_x"), (sigma_x, n * n, "sigma_x")):
        if t.numel() not in (per * T * N, per * (T + 1) * N) or (t.numel() != per * T * N and N != 1 and
                                                                 layout != LAYOUT_PROBLEM_MINOR):
            raise ValueError("%s must hold %d or %d slices (the latter only in PROBLEM_MINOR)" % (nm, T, T + 1))
        _chk(t, t.numel(), nm)
    dev = p_K.device
    kl = empty((N, T), layout, dev)
    status = torch.empty(N, dtype=torch.int32, device=dev)
    check(lib().trajopt_gps_kl_divergence_batched(
        *map(dptr, (p_K, p_kff, p_sigma_ctl, q_K, q_kff, q_sigma_ctl, mu_x, sigma_x)), n, m, T, C.c_int64(N),
        int(layout), dptr(kl), dptr(status), stream_ptr()))
    return kl, status


def gps_quad_expectation(mu, sigma_s, Q, q, q0, dm_state, nb_problems, layout=LAYOUT_PROBLEM_MINOR):
    """gps/src/util.cpp:123-134 for a batch -> [N] tensor.  mu, q: (N, n); sigma_s, Q: (N, n, n); q0: [N]."""
    import torch
    _lib.require_gpu()
    n, N = dm_state, nb_problems
    _gps_check(((mu, n, "mu"), (sigma_s, n * n, "sigma_s"), (Q, n * n, "Q"), (q, n, "q"), (q0, 1, "q0")), N)
    out = torch.empty(N, dtype=torch.float64, device=mu.device)
    check(lib().trajopt_gps_quad_expectation_batched(*map(dptr, (mu, sigma_s, Q, q, q0)), n, C.c_int64(N),
                                                     int(layout), dptr(out), stream_ptr()))
    return out


def bsp_backward_pass(Q, q, R, r, P, p, F, G, T_, U, V, X, Y, Z, lmbda, reg, dm_belief, dm_act, nb_steps, nb_problems,
                      layout=LAYOUT_PROBLEM_MINOR, tau_mode=0):
    """Belief-space iLQR backward pass of /root/reference/trajopt/bspilqr/src/util.cpp:83-204 for a batch.
    Device tensors in PROBLEM_MINOR; Q, q, p hold nb_steps + 1 slices, the others nb_steps (or more: a prefix is read);
    lmbda is a [N] tensor.  Returns S, s, tau, dS (N, 2), K, kff, diverge (-1 = completed)."""
    import torch
    _lib.require_gpu()
    b, m, T, N = dm_belief, dm_act, nb_steps, nb_problems
    bb = b * b
    for t, cnt, nm in ((Q, bb * (T + 1), "Q"), (q, b * (T + 1), "q"), (p, bb * (T + 1), "p")):
        _chk(t, cnt * N, nm)
    for t, per, nm in ((R, m * m, "R"), (r, m, "r"), (P, b * m, "P"), (F, bb, "F"), (G, b * m, "G"), (T_, bb * b, "T"),
                       (U, bb * bb, "U"), (V, bb * m, "V"), (X, bb * b, "X"), (Y, bb * bb, "Y"), (Z, bb * m, "Z")):
        if t.numel() < per * T * N:
            raise ValueError("%s holds fewer than %d slices" % (nm, T))
        _chk(t, t.numel(), nm)
    _chk(lmbda, N, "lmbda")
    dev = Q.device
    o = dict(S=empty((N, b, b, T + 1), layout, dev, zero=True), s=empty((N, b, T + 1), layout, dev, zero=True),
             tau=empty((N, bb, T + 1), layout, dev, zero=True), dS=empty((N, 2), layout, dev, zero=True),
             K=empty((N, m, b, T), layout, dev, zero=True), kff=empty((N, m, T), layout, dev, zero=True),
             diverge=torch.empty(N, dtype=torch.int32, device=dev))
    io = BspBackwardIO()
    for k, t in zip(("Q", "q", "R", "r", "P", "p", "F", "G", "T", "U", "V", "X", "Y", "Z", "lmbda"),
                    (Q, q, R, r, P, p, F, G, T_, U, V, X, Y, Z, lmbda)):
        setattr(io, k, dptr(t))
    for k in ("S", "s", "tau", "dS", "K", "kff", "diverge"):
        setattr(io, k, dptr(o[k]))
    check(lib().trajopt_bspilqr_backward_pass_batched(C.byref(io), int(reg), int(tau_mode), b, m, T, C.c_int64(N),
                                                      int(layout), stream_ptr()))
    return o
