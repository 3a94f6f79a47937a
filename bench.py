#!/usr/bin/env python3
"""bench.py - problem-timesteps/sec of one batched iLQR iteration on MI355X (BASELINE.json metric).

A "step" is ONE full batched iLQR iteration over the whole batch, exactly the body of the reference's
`for iter in range(nb_iter)` (/root/reference/trajopt/ilqr/ilqr.py:151-231): quadratise the dynamics and the
cost along the current trajectory -> regularised Riccati backward pass (lambda retries inside the kernel)
-> forward rollout + cost for all 11 alphas -> first-accept line search, accept/reject, lambda update.
Headline workload = BASELINE config C2: cartpole (dx=4, du=1), T=500, 4096 instances per GPU, fp64, synthetic
initial states x0 + N(0, sigma0) (seed 1337), resident in HBM before the timed region.  The stopping rules
(tolfun / tolgrad / max_lmbda) are disabled so that every problem does a full iteration in every timed step.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

The same run also times, each in its own bracketed region after the headline, the other workloads BASELINE.json
names (key `configs` of the JSON line; --no-configs skips them):
  cartpole_65536   north_star's target: 65 536 cartpoles on ONE GPU, full iteration
  c3_share         C3: 8192 double cartpoles (one GPU's share of 65 536 over 8), full iteration - with --gpus N
                   every rank runs its own share, i.e. N x 8192 problems sharded with no collective
  c4_quadrotor     C4: 4096 quadrotors, the 8-alpha line search, full iteration
  c5_riccati       C5: TV-LQR Riccati sweep, 1024 x (64 + 16), T = 1000
  backward_dense   the backward pass alone on dense synthetic inputs at C2 / C3 / C4 / 65 536 cartpoles,
                   gains-only and full-API outputs, against the HBM roofline

Multi-GPU: problems are independent, so each rank owns its own instances (weak scaling) and there is no data-path
collective; RCCL is used for the barrier and to take the max time over ranks.
Rank 0 prints one JSON line (see DESIGN.md "Measurement" for the roofline / cpu_baseline definitions).
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
# fp64 VECTOR peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s (AMD's MI355X figure; the guide
# lists matrix-core peaks only).  The rollout issues plain v_fma_f64 / v_mul_f64 / v_add_f64: this is its ceiling.
# The fp64 matrix cores have the same peak (a v_mfma_f64_16x16x4 = 2048 flop is busy for 64 cycles of one SIMD).
FP64_VALU_PEAK_TFLOPS = 78.6
FP64_MFMA_PEAK_TFLOPS = 78.6
PHASES = ("prep", "quadratize", "backward", "post_backward", "rollout", "select_commit")
KERNEL_PHASES = ("quadratize", "backward", "rollout", "select_commit")
ENV_DEFAULTS = {"cartpole": (0.02, 1e-5), "pendulum": (0.05, 1e-5), "double_cartpole": (0.01, 1e-5),
                "quadrotor": (0.02, 1e-3)}
CENSUS_FILE = "r03_rollout_isa_census.json"
ALL_ALPHAS = np.power(10., np.linspace(0, -3, 11))   # ilqr.py:15


def alg_bytes_per_problem_step(n, m):
    """ALGORITHMIC HBM bytes per problem-timestep of each kernel (fp64; every array touched once), SURVEY 8(d).
    backward: gains-only form, dense inputs (backward_full: the full-API form).  quadratize: reads xref, uref; writes
    A, B, Cxx, cx, Cuu, cu, Cxu.  rollout: reads K, kff, xref, uref once; writes the state/action of ONE (the accepted)
    candidate - the other candidates' stores are a design cost of the parallel line search, not algorithmic bytes."""
    return {
        "backward": 8 * (2 * n * n + 3 * n * m + m * m + n + 2 * m),
        "backward_full": 8 * (4 * n * n + 4 * n * m + 2 * m * m + 3 * n + 3 * m),
        "quadratize": 8 * (2 * n * n + 2 * n * m + m * m + 2 * n + 2 * m),
        "rollout": 8 * (n * m + 2 * m + n) + 8 * (n + m),
        "select_commit": 2 * 8 * (n + m + n * m + m),
    }


def moved_bytes_per_problem_step(n, m, plan):
    """Bytes the LAUNCHED variant of each phase moves per problem-timestep (what `frac` of the HBM roofline is based
    on; the structured cost and the fused kernel skip most of the dense algorithmic bytes above).
    fused quadratise + backward: reads x, u through ref_plane, writes them through to xref / uref, writes K, kff.
    structured two-kernel path: quadratise reads x, u, writes A, B, cx, cu (+ x, u through); backward reads A, B, cx,
    cu, uref, writes K, kff.  dense: the algorithmic bytes."""
    xu, gains = 8 * (n + m), 8 * (n * m + m)
    if plan["fused_quad_bwd"]:
        return {"backward": 2 * xu + gains, "quadratize": 0}
    if plan["structured_cost"]:
        lin = 8 * (n * n + n * m + n + m)
        return {"backward": lin + 8 * m + gains, "quadratize": 2 * xu + lin}
    a = alg_bytes_per_problem_step(n, m)
    return {"backward": a["backward"], "quadratize": a["quadratize"] + xu}


def stamped(path, stamp):
    """JSON kept under profiles/ if it was made from the current kernel sources (its "stamp"), else None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", path)))
    except (OSError, ValueError):
        return None
    return d if d.get("stamp") == stamp else None


def rollout_census():
    """The rollout's ISA census (scripts/rollout_census.py) if it was taken on the current rollout sources."""
    from trajopt_amd import _lib
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", CENSUS_FILE)))
    except (OSError, ValueError):
        return None
    return d if d.get("rollout_stamp") == _lib.source_stamp(_lib.ROLLOUT_SOURCES) else None


def rounded(x, sig=6):
    """The extra (non-contract) parts of the line, rounded so that the line stays short enough to survive a tail."""
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else x
    if isinstance(x, dict):
        return {k: rounded(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [rounded(v, sig) for v in x]
    return x


def make_problem_set(env, N, T, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    x0 = np.asarray(env.x0)[None] + rng.multivariate_normal(np.zeros(env.dm_state), env.sigma0, size=N)
    kff0 = 1e-4 * rng.standard_normal((N, env.dm_act, T))
    return x0, kff0


def host_cores():
    """(cores the machine has, cores this process may run on, cgroup CPU quota in cores or None)."""
    total = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = total
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return total, usable, quota


def cpu_baseline(env_name, dt, uw, T, seed, budget_s=12.0):
    """The CPU oracle's iLQR (the dependency-free C++ restatement of util.cpp + ilqr.py; kind "port": the
    reference's Armadillo/autograd path cannot be built here) on a bounded sample of the same workload, one
    problem per thread (the reference's own fan-out is one problem per process, examples/ilqr/mpc/pendulum_parallel.py:40-45).
    Timed twice: on the GPU box's CPU share (OMP_NUM_THREADS, 16 there) and on EVERY core the process may run on;
    `value` / `cores` are the faster of the two - the baseline the GPU number is divided by."""
    from oracle import oracle as O
    from trajopt_amd import envs
    total, usable, quota = host_cores()
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(usable, 16)
    share = max(1, min(share, usable))
    oe = O.make_env(env_name, dt=dt, uw=[uw])
    pe = envs.make(env_name, dt=dt, uw=uw)
    iters = 4
    opts = [1., 1., 1e-6, float("inf"), 1.6, -float("inf"), 0.0, 0.]   # stopping disabled, as the GPU run

    def timed(threads, budget, salt):
        done_steps, elapsed, problems, chunk = 0, 0.0, 0, 4 * threads
        while elapsed < budget and problems < 262144:
            x0, kff0 = make_problem_set(pe, chunk, T, seed + salt + problems)
            sec, _, it = O.bench_ilqr(oe, T, x0, kff0.reshape(chunk, -1), iters, threads, opts=opts)
            elapsed += sec
            done_steps += it * T
            problems += chunk
        return {"value": done_steps / elapsed, "threads": threads, "problems": problems, "seconds": elapsed}

    runs = [timed(share, budget_s * (0.5 if usable > share else 1.0), 7919)]
    if usable > share:   # every core of the host (bounded: 512 threads at most)
        runs.append(timed(min(usable, 512), budget_s * 0.5, 15485863))
    best = max(runs, key=lambda r: r["value"])
    out = {"value": best["value"], "unit": "problem-timesteps/s", "cores": best["threads"], "kind": "port",
           "sample": "%d %s problems x T=%d x %d iLQR iterations (sequential first-accept line search), "
                     "%.1f s on %d threads, oracle/liboracle.so -O3" % (best["problems"], env_name, T, iters, best["seconds"], best["threads"]),
           "host_cores_total": total, "host_cores_usable": usable, "cgroup_cpu_quota": quota,
           "runs": [{"threads": r["threads"], "value": r["value"], "problems": r["problems"], "seconds": r["seconds"]} for r in runs]}
    # the same port on ONE core, and the NumPy mirror as an indication of the Python-level cost the
    # reference pays per problem (its own autograd path is not runnable here and was not measured)
    x0, kff0 = make_problem_set(pe, 64, T, seed + 104729)
    sec, _, it = O.bench_ilqr(oe, T, x0, kff0.reshape(64, -1), iters, 1, opts=opts)
    out["one_core"] = {"value": it * T / sec, "sample": "64 problems x %d iterations, %.1f s" % (iters, sec)}
    try:
        from oracle import numpy_mirror as Mi
        me = Mi.Env(env_name, dt=dt, uw=[uw])
        Tm, reps, secm = min(T, 50), 0, 0.0
        while secm < 1.5 and reps < 64:   # at least a second of work: one problem, one iteration at a time
            t0 = time.perf_counter()
            Mi.ilqr_run(me, Tm, x0[reps], kff0[reps][:, :Tm], np.ones(Tm + 1), O.DEFAULT_ALPHAS, 1)
            secm += time.perf_counter() - t0
            reps += 1
        out["numpy_mirror"] = {"value": reps * Tm / secm, "sample": "%d problems x T=%d x 1 iteration, %.1f s, 1 core" % (reps, Tm, secm)}
    except Exception as exc:   # the mirror is an extra, never a reason to lose the bench line
        out["numpy_mirror"] = {"value": None, "sample": "failed: %r" % (exc,)}
    return out


class Ctx:
    """What every timed line needs: torch, the library, the rank fence and the max over ranks."""

    def __init__(self, torch, L, dist, red_dev, rank, world):
        self.torch, self.L, self.dist, self.red_dev, self.rank, self.world = torch, L, dist, red_dev, rank, world

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        tt = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.red_dev)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def release(self):
        gc.collect()
        self.torch.cuda.empty_cache()


def iteration_line(cx, env_name, N, T, alphas, K, W, seed, staged=-1, no_fusion=False, max_lmbda=float("inf"), sub_batches=1):
    """W untimed + K timed iLQR iterations of N problems on this rank -> measurements (all ranks take part in the
    fences; the elapsed time is the max over ranks).
    sub_batches: 1 = one batch on one stream; S > 1 (0: the solver's own rule, trajopt_amd/ilqr/streamed.py) = after
    that measurement the SAME problems once more as S sub-batches on streams of their own (StreamedILQR), W untimed +
    K timed iterations -> out["streamed"]; phases and the dominant kernel's time always come from the one-batch run
    (launches of different sub-batches overlap: per-kernel times are not separable there).
    max_lmbda: inf = no problem ever gives up (the headline's setting: constant work per step); the reference's
    default 1e6 lets a problem whose backward pass cannot be regularised stop as ilqr.py:168,228 does - the line then
    counts ONLY the problem-iterations that were executed (`problem_iters`)."""
    from trajopt_amd import _lib, envs
    from trajopt_amd.ilqr import BatchedILQR
    torch, L = cx.torch, cx.L
    dt, uw = ENV_DEFAULTS[env_name]
    env = envs.make(env_name, dt=dt, uw=uw)
    n, m = env.dm_state, env.dm_act
    x0, kff0 = make_problem_set(env, N, T, seed)
    skw = dict(action_penalty=None, apply_env_kwargs=False, alphas=alphas, tolfun=-float("inf"), tolgrad=0.0,
               max_lmbda=max_lmbda, staged_line_search=staged, no_fusion=no_fusion, trace_cap=W + K + 16)
    solver = BatchedILQR(env, T, x0, kff_init=kff0, **skw)
    del x0, kff0

    def profiled(iters, mask=None):
        """`iters` iterations with hipEvents around the phases in `mask` (None: all six) -> mean ms per phase."""
        _lib.check(L.trajopt_profile_begin(iters))
        if mask is not None:
            _lib.check(L.trajopt_profile_select(C.c_uint(mask)))
        solver.iterate(iters)
        torch.cuda.synchronize()
        ms = (C.c_double * len(PHASES))()
        nrec = C.c_int(0)
        _lib.check(L.trajopt_profile_end(ms, C.byref(nrec)))
        return {k: ms[i] / max(nrec.value, 1) for i, k in enumerate(PHASES)}

    # An event costs the stream about 5 us here (all seven boundaries of an iteration: 0.926 vs 0.888 ms), so the
    # timed region brackets ONE phase - the kernel with the largest device time, found in the (fully bracketed)
    # warmup - and the six-phase breakdown is taken in a pass of its own after the timed region.
    solver.init_rollout()
    dom = "rollout"
    if W >= 2:   # the first iteration carries the cold start of every kernel: it is never the one that is bracketed
        solver.iterate(W - min(2, W - 1))
        warm = profiled(min(2, W - 1))
        dom = max(KERNEL_PHASES, key=lambda k: warm[k])
    else:
        solver.iterate(W)

    iters0 = solver.host("iters").astype(np.int64)
    cx.fence()
    _lib.check(L.trajopt_profile_begin(K))
    _lib.check(L.trajopt_profile_select(C.c_uint(1 << PHASES.index(dom))))
    t0 = time.perf_counter()
    solver.iterate(K)
    cx.fence()
    t1 = time.perf_counter()
    phase_ms = (C.c_double * len(PHASES))()
    nrec = C.c_int(0)
    _lib.check(L.trajopt_profile_end(phase_ms, C.byref(nrec)))
    dom_ms = phase_ms[PHASES.index(dom)] / max(nrec.value, 1)
    iters1 = solver.host("iters").astype(np.int64)
    KB = min(K, 10)
    phases = profiled(KB)
    elapsed = cx.max_over_ranks(t1 - t0)

    status = solver.host("status")
    iters = solver.host("iters")
    ret = solver.host("last_return")
    if max_lmbda == float("inf"):
        assert np.all(status == 0) and np.all(iters == W + K + KB), "a problem stopped inside the timed region"
        assert np.all(np.isfinite(ret)), "non-finite trajectory cost"
    else:
        assert np.all((status == 0) | (status == 3)), "only the max_lmbda rule is armed"
    out = {"env": env_name, "n": n, "m": m, "N": N, "T": T, "nb_alphas": int(np.asarray(alphas).size), "K": K, "W": W, "KB": KB,
           "elapsed": elapsed, "dom": dom, "dom_ms": dom_ms, "phases": phases, "plan": solver.plan(),
           "problem_iters": int((iters1 - iters0).sum()), "gave_up": int((status == 3).sum()),
           "mean_return": float(ret[status == 0].mean())}
    del solver
    cx.release()
    from trajopt_amd.ilqr import StreamedILQR, auto_sub_batches
    S = auto_sub_batches(n, m, N) if sub_batches == 0 else int(sub_batches)
    if S > 1:
        x0, kff0 = make_problem_set(env, N, T, seed)
        sub = StreamedILQR(env, T, x0, kff_init=kff0, sub_batches=S, **skw)
        del x0, kff0
        sub.init_rollout()
        sub.iterate(W)
        iters0 = sub.host("iters").astype(np.int64)
        cx.fence()
        t0 = time.perf_counter()
        sub.iterate(K)
        cx.fence()
        t1 = time.perf_counter()
        iters1, status, ret = sub.host("iters").astype(np.int64), sub.host("status"), sub.host("last_return")
        if max_lmbda == float("inf"):
            assert np.all(status == 0) and np.all(iters1 == W + K), "a problem stopped inside the timed region"
            assert np.all(np.isfinite(ret)), "non-finite trajectory cost"
        out["streamed"] = {"sub_batches": sub.sub_batches, "elapsed": cx.max_over_ranks(t1 - t0), "problem_iters": int((iters1 - iters0).sum()),
                           "gave_up": int((status == 3).sum()), "mean_return": float(ret[status == 0].mean())}
        del sub
        cx.release()
    return out


def roofline_of(r, stamp, pmc=None, sq=None):
    """Roofline of the dominant kernel of an iteration line."""
    n, m, N, T, dom, dom_ms, plan = r["n"], r["m"], r["N"], r["T"], r["dom"], r["dom_ms"], r["plan"]
    ab = alg_bytes_per_problem_step(n, m)
    mb = moved_bytes_per_problem_step(n, m, plan)
    census = rollout_census()
    traffic = pmc["by_phase"].get(dom) if pmc else None
    hbm_alg = ab[dom] * N * T / (dom_ms * 1e-3) / 1e9
    two_pass = plan["staged_line_search"] == 2
    ck = census["kernels"].get(r["env"]) if census else None
    if dom == "rollout" and ck:
        # the rollout is a chain of plain fp64 vector instructions: priced against the fp64 VALU peak, with the
        # flops of one step counted in the kernel's ISA (scripts/rollout_census.py)
        fl = ck["fp64_flops_per_lane_step"]
        # two-pass line search (plan 2): every (problem, alpha) once for its return, the accepted alpha of each problem
        # once more for its trajectory - the phase holds both launches and the select kernel between them
        lanes = N * r["nb_alphas"] + (N if two_pass else 0)
        achieved = fl * lanes * T / (dom_ms * 1e-3) / 1e12
        return {"bound": "fp64-valu", "kernel": dom, "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP64_VALU_PEAK_TFLOPS, "traffic": traffic,
                "fp64_flops_per_lane_step": fl,
                # counted in the straight-line form's block loop; the rolled / returns-only forms (oversubscribed
                # batches) run the same fp64 arithmetic per step around different loop / addressing code
                "instructions_per_step": ck["instructions_per_step"] if plan["rollout_straight_line"] and not two_pass else None,
                "flops_counted_on": "this kernel" if plan["rollout_straight_line"] and not two_pass else "the straight-line form of the same model (same arithmetic per step)",
                "lanes": lanes, "waves_per_simd": lanes / 64.0 / 1024.0,
                "launches": "returns of every alpha, select, the accepted alpha again" if two_pass else "one",
                "hbm_algorithmic_GBps": hbm_alg, "hbm_frac": hbm_alg / HBM_PEAK_GBS,
                "alg_bytes_per_launch": ab[dom] * N * T, "avg_launch_ms": dom_ms,
                "valu_insts_per_simd_cycle_pmc": (sq["kernels_by_phase"].get(dom, {}).get("valu_insts_per_simd_cycle") if sq else None)}
    moved = mb.get(dom, ab[dom]) * N * T
    achieved = moved / (dom_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "moved_bytes_per_launch": moved,
            "alg_bytes_per_launch": ab[dom] * N * T, "avg_launch_ms": dom_ms}


def backward_of(r):
    mb = moved_bytes_per_problem_step(r["n"], r["m"], r["plan"])
    ab = alg_bytes_per_problem_step(r["n"], r["m"])
    g = mb["backward"] * r["N"] * r["T"] / (r["phases"]["backward"] * 1e-3) / 1e9
    return {"moved_bytes_per_problem_step": mb["backward"], "GBps_moved": g, "frac_of_hbm_peak": g / HBM_PEAK_GBS,
            "dense_algorithmic_bytes_per_problem_step": ab["backward"],
            "note": "quadratise + backward in one launch" if r["plan"]["fused_quad_bwd"] else "backward kernel alone"}


def config_line(r, world, stamp, label):
    """One entry of `configs` for an iteration line."""
    roof = roofline_of(r, stamp)
    keep = ("bound", "kernel", "achieved", "unit", "frac", "avg_launch_ms")
    t = r.get("streamed", r)   # the timed run that counts: the sub-batched one where the solver's rule takes it
    out = {"workload": label, "n_gpus": world, "steps": r["K"], "ms": t["elapsed"] / r["K"] * 1e3,
           # problem-timesteps/s over the problem-iterations actually executed in the timed region (= N K unless a problem gave up)
           "value": world * t["problem_iters"] * r["T"] / t["elapsed"],
           "roofline": {k: roof[k] for k in keep if k in roof},
           "phase_ms": {k: r["phases"][k] for k in KERNEL_PHASES},
           "backward_frac_of_hbm": backward_of(r)["frac_of_hbm_peak"],
           "plan": {k: v for k, v in r["plan"].items() if k != "pad_"}}
    if "streamed" in r:
        out["sub_batches"] = t["sub_batches"]
        out["one_batch"] = {"ms": r["elapsed"] / r["K"] * 1e3, "value": world * r["problem_iters"] * r["T"] / r["elapsed"],
                            "note": "the same problems and iterations as ONE batch on one stream; roofline and phase_ms are "
                                    "taken on this run (launches of different sub-batches overlap)"}
    if t["gave_up"]:
        out["problem_iters"], out["gave_up"] = t["problem_iters"], t["gave_up"]
    return out


def timed_calls(torch, call, reps):
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def backward_dense_line(cx, n, m, T, N, layout, reps=5):
    """The backward pass alone (util.cpp:83-182) on dense synthetic inputs resident in HBM, lambda = 1, reg = 1:
    gains-only (what an iteration needs) and full-API outputs, against the HBM roofline on ALGORITHMIC bytes."""
    from trajopt_amd import backward
    from trajopt_amd.synthetic import device_backward_inputs
    torch = cx.torch
    d = device_backward_inputs(n, m, T, N, layout, affine=False)
    lam = torch.ones(N, dtype=torch.float64, device="cuda")
    ab = alg_bytes_per_problem_step(n, m)
    res = {"shape": "dx=%d du=%d T=%d N=%d" % (n, m, T, N)}
    for key, full in (("gains_only", False), ("full_api", True)):
        out = {}
        call = lambda: backward.ilqr_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], lam, 1, n, m,
                                                   T, N, layout, full=full, out=out)
        ms = timed_calls(torch, call, reps)
        assert int((out["diverge"] >= 0).sum()) == 0 and bool(torch.isfinite(out["K"]).all())
        b = ab["backward_full" if full else "backward"]
        g = N * T * b / ms / 1e6
        res[key] = {"ms": ms, "bytes_per_problem_step": b, "GBps": g, "frac": g / HBM_PEAK_GBS}
        del out
    del d, lam
    cx.release()
    return res


def riccati_line(cx, n=64, m=16, T=1000, N=1024, reps=3):
    """C5: the TV-LQR sweep (riccati.py:100-126) on dense synthetic inputs in HBM."""
    from trajopt_amd import backward
    from trajopt_amd.synthetic import device_backward_inputs
    torch = cx.torch
    d = device_backward_inputs(n, m, T, N, 0, affine=True)
    keep = {}

    def call():   # the outputs (42 GB at C5) are allocated by the first call and written again by the others
        keep["o"] = backward.riccati_backward_pass(d["Cxx"], d["cx"], d["Cuu"], d["cu"], d["Cxu"], d["A"], d["B"], d["c"], n, m, T, N, 0,
                                                   out=keep.get("o"))

    ms = timed_calls(torch, call, reps)
    o = keep["o"]
    assert int((o["status"] != 0).sum()) == 0 and bool(torch.isfinite(o["K"]).all())
    b = 8 * (3 * n * n + 3 * n * m + m * m + 4 * n + 2 * m)   # in A,B,c,Cxx,Cuu,Cxu,cx,cu; out K,kff,V,v
    fl = 4 * n ** 3 + 10 * n * n * m + 6 * n * m * m + 2 * m ** 3 + 2 * n * n
    tf, g = N * T * fl / ms / 1e9, N * T * b / ms / 1e6
    del d, o, keep
    cx.release()
    return {"workload": "C5: TV-LQR Riccati sweep dx=%d du=%d T=%d x %d problems, inputs and outputs resident in HBM" % (n, m, T, N),
            "ms": ms, "value": N * T / ms * 1e3,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                         "flops_per_problem_step": fl, "hbm_GBps": g, "hbm_frac": g / HBM_PEAK_GBS, "bytes_per_problem_step": b}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--instances", type=int, default=4096, help="problems per GPU (C2: 4096)")
    ap.add_argument("--horizon", type=int, default=500)
    ap.add_argument("--env", default="cartpole")
    ap.add_argument("--alphas", type=int, default=11, help="the first k of the reference's 11 step sizes (C4: 8)")
    ap.add_argument("--staged", type=int, default=-1, help="line search: -1 auto, 0 one launch, 1 staged, 2 returns first then the accepted alpha again (same results)")
    ap.add_argument("--no-fusion", action="store_true", help="quadratise and backward pass as two launches (same results)")
    ap.add_argument("--max-lmbda", type=float, default=float("inf"),
                    help="headline line only: the reference's give-up rule (ilqr.py:17 uses 1e6); inf = every problem iterates in every step")
    ap.add_argument("--sub-batches", type=int, default=1,
                    help="headline line: 1 = one batch on one stream (default); S > 1 = the problems as S sub-batches on HIP streams of "
                         "their own (StreamedILQR); 0 = the solver's rule (1 at C2).  The configs block always takes the rule")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="the headline line only")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # one process per GPU; TRAJOPT_BENCH_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the
    # multi-rank path on a single-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("TRAJOPT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local >= ndev:
        raise SystemExit("rank %d has no GPU (%d visible)" % (local, ndev))
    torch.cuda.set_device(local % ndev)
    dist = None
    # TRAJOPT_BENCH_FORCE_DIST=1: take the collective path with ONE rank too (tests: RCCL at world size 1 on the device)
    if world > 1 or os.environ.get("TRAJOPT_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    devices = None
    if dist is not None:
        # every rank must own a GPU of its own: gather (device index, uuid) and compare
        props = torch.cuda.get_device_properties(local % ndev)
        me = "%d:%s" % (local % ndev, getattr(props, "uuid", "") or getattr(props, "pci_bus_id", ""))
        devices = [None] * dist.get_world_size()
        dist.all_gather_object(devices, me)
        if backend == "nccl":
            assert len(set(devices)) == dist.get_world_size(), "ranks share a device: %s" % (devices,)

    from trajopt_amd import _lib

    N, T, K, W = a.instances, a.horizon, a.steps, a.warmup
    alphas = ALL_ALPHAS[:a.alphas]
    cx = Ctx(torch, _lib.lib(), dist, red_dev, rank, world)
    r = iteration_line(cx, a.env, N, T, alphas, K, W, 1337 + rank, staged=a.staged, no_fusion=a.no_fusion, max_lmbda=a.max_lmbda,
                       sub_batches=a.sub_batches)
    stamp = _lib.source_stamp()
    n, m = r["n"], r["m"]

    configs = {}
    if not a.no_configs:
        # C3: every rank its share of 8192 double cartpoles (world x 8192 problems, no collective) - at any world size
        c3 = iteration_line(cx, "double_cartpole", 8192, 500, ALL_ALPHAS, 10, 3, 4001 + rank, sub_batches=0)
        configs["c3_share"] = config_line(c3, world, stamp, "C3: 8192 double cartpoles per GPU (dx=6 du=1 T=500, 11 alphas), %d GPU(s)" % world)
        if world == 1:   # single-GPU workloads
            q = iteration_line(cx, "cartpole", 65536, 500, ALL_ALPHAS, 10, 3, 2001, sub_batches=0)
            configs["cartpole_65536"] = config_line(q, 1, stamp, "north_star target: 65 536 cartpoles on one GPU (dx=4 du=1 T=500, 11 alphas)")
            # C4 twice.  The start (free fall under kff = 1e-4 randn, ilqr.py:70) sends ~4 % of the quadrotors tumbling
            # through the Euler-angle singularity; their backward pass overflows for every lambda the reference allows.
            # (a) max_lmbda = 1e6, the reference's default: such a problem gives up as ilqr.py:168,228 says and only
            # executed problem-iterations count; (b) nothing ever gives up (the headline's rule): each of them then
            # takes ~54 growing-lambda sweeps per iteration until lambda overflows, and the launch waits for them.
            q = iteration_line(cx, "quadrotor", 4096, 500, ALL_ALPHAS[:8], 8, 3, 3001, max_lmbda=1e6, sub_batches=0)
            configs["c4_quadrotor"] = config_line(q, 1, stamp, "C4: 4096 quadrotors (dx=12 du=4 T=500), 8 alphas, max_lmbda=1e6 (reference default): hopeless problems give up, only executed problem-iterations counted")
            q = iteration_line(cx, "quadrotor", 4096, 500, ALL_ALPHAS[:8], 8, 3, 3001, sub_batches=0)
            configs["c4_quadrotor_never_give_up"] = config_line(q, 1, stamp, "C4 with max_lmbda=inf: every problem iterates in every step")
            configs["c5_riccati"] = riccati_line(cx)
            configs["backward_dense"] = {
                "note": "backward pass alone, dense synthetic inputs in HBM, lambda=1 reg=1; roofline per entry: bound hbm, GBps = SURVEY 8(d) "
                        "algorithmic bytes / time, frac = GBps / 8000",
                "c2_4096": backward_dense_line(cx, 4, 1, 500, 4096, 1),
                "c3_8192": backward_dense_line(cx, 6, 1, 500, 8192, 1),
                "c4_4096": backward_dense_line(cx, 12, 4, 500, 4096, 0),
                "cartpole_65536": backward_dense_line(cx, 4, 1, 500, 65536, 1),
                "double_cartpole_65536": backward_dense_line(cx, 6, 1, 500, 65536, 1)}

    if rank == 0:
        nb_alphas = r["nb_alphas"]
        is_c2 = a.env == "cartpole" and N == 4096 and T == 500 and nb_alphas == 11
        tag = {("cartpole", 4096, 500): "C2", ("double_cartpole", 8192, 500): "C3 (one GPU's share)",
               ("quadrotor", 4096, 500): "C4", ("pendulum", 1, 100): "C1"}.get((a.env, N, T), "custom")
        # HBM traffic / issue counters of the kernels from the committed rocprofv3 --pmc passes (separate runs, gfx950
        # read correction applied: scripts/pmc_traffic.py, scripts/pmc_sq.py) - PMC counters cannot be sampled from
        # inside this process, so they are quoted only for the workload they were collected on AND only while the
        # kernel sources are the ones they were collected from (stamp)
        pmc = stamped("r03_pmc_traffic_c2.json", stamp) if is_c2 else None
        sq = stamped("r03_pmc_sq_c2.json", stamp) if is_c2 else None
        timed = r.get("streamed", r)
        out = {
            "metric": "problem-timesteps/sec (iLQR iteration, batched)",
            "value": world * N * T * K / timed["elapsed"],
            "unit": "problem-timesteps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": timed["elapsed"] / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: batched iLQR %s dx=%d du=%d T=%d, %d instances per GPU, %d alphas, reg=1, "
                                   "stopping rules disabled" % (tag, a.env, n, m, T, N, nb_alphas),
                       "instances_per_gpu": N, "horizon": T, "parallelism": "problems sharded x%d, no collective" % world},
            "roofline": rounded(roofline_of(r, stamp, pmc, sq)),
            "phase_ms": rounded(r["phases"]),
            "phase_ms_note": "%d iterations bracketed after the timed region (which carries only the dominant kernel's events)" % r["KB"],
            "plan": r["plan"],
            "backward_pass": rounded(backward_of(r)),
            "source_stamp": stamp,
            "mean_return": r["mean_return"],
        }
        if "streamed" in r:
            out["sub_batches"] = timed["sub_batches"]
            out["one_batch_ms_per_step"] = r["elapsed"] / K * 1e3
            out["sub_batches_note"] = ("ms_per_step / value: the problems as sub-batches on HIP streams of their own; roofline, phase_ms and "
                                       "one_batch_ms_per_step: the same problems and iterations as one batch on one stream")
        if devices is not None:
            out["rank_devices"] = devices
        if not a.no_cpu_baseline and world == 1:   # (the CPU baseline is a rank-0, N = 1 leg: a scaling run does not repeat it)
            dt, uw = ENV_DEFAULTS[a.env]
            cb = cpu_baseline(a.env, dt, uw, T, 1337, a.cpu_budget)
            out["cpu_baseline"] = rounded(cb)
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
            q = configs.get("cartpole_65536")
            if q is not None and a.env == "cartpole" and T == 500:
                # the CPU port's rate per problem does not depend on how many problems are queued (one per thread), so
                # the cartpole baseline above is also the 65 536-instance baseline
                q["speedup_vs_cpu_baseline"] = q["value"] / cb["value"]
                q["cpu_baseline_cores"] = cb["cores"]
        if configs:
            out["configs"] = rounded(configs, 5)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
