#!/usr/bin/env python3
"""bench.py - problem-timesteps/sec of one batched iLQR iteration on MI355X (BASELINE.json metric).

A "step" is ONE full batched iLQR iteration over the whole batch, exactly the body of the reference's
`for iter in range(nb_iter)` (/root/reference/trajopt/ilqr/ilqr.py:151-231): quadratise the dynamics and the
cost along the current trajectory -> regularised Riccati backward pass (lambda retries inside the kernel)
-> forward rollout + cost for all 11 alphas -> first-accept line search, accept/reject, lambda update.
Workload = BASELINE config C2: cartpole (dx=4, du=1), T=500, 4096 instances per GPU, fp64, synthetic
initial states x0 + N(0, sigma0) (seed 1337), resident in HBM before the timed region.  The stopping rules
(tolfun / tolgrad / max_lmbda) are disabled so that every problem does a full iteration in every timed step.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Multi-GPU: problems are independent, so each rank owns its own 4096 instances (weak scaling) and there is
no data-path collective; RCCL is used for the barrier and to take the max time over ranks.
Rank 0 prints one JSON line (see DESIGN.md "Measurement" for the roofline / cpu_baseline definitions).
"""
import argparse
import ctypes as C
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
FP64_VALU_PEAK_TFLOPS = 78.6
PHASES = ("prep", "quadratize", "backward", "post_backward", "rollout", "select_commit")


def alg_bytes_per_problem_step(n, m, nb_alphas):
    """ALGORITHMIC HBM bytes per problem-timestep of each kernel (fp64; every array touched once), SURVEY 8(d).
    backward: gains-only form, dense inputs.  quadratize: reads xref, uref; writes A, B, Cxx, cx, Cuu, cu, Cxu.
    rollout: reads K, kff, xref, uref once; writes the state/action of ONE (the accepted) candidate - the other
    candidates' stores are a design cost of the parallel line search, not algorithmic bytes."""
    return {
        "backward": 8 * (2 * n * n + 3 * n * m + m * m + n + 2 * m),
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
    a = alg_bytes_per_problem_step(n, m, 0)
    return {"backward": a["backward"], "quadratize": a["quadratize"] + xu}


def stamped(path, stamp):
    """JSON kept under profiles/ if it was made from the current kernel sources (its "stamp"), else None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", path)))
    except (OSError, ValueError):
        return None
    return d if d.get("stamp") == stamp else None


def make_problem_set(env, N, T, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    x0 = np.asarray(env.x0)[None] + rng.multivariate_normal(np.zeros(env.dm_state), env.sigma0, size=N)
    kff0 = 1e-4 * rng.standard_normal((N, env.dm_act, T))
    return x0, kff0


def cpu_baseline(env_name, dt, uw, T, seed, budget_s=12.0):
    """The CPU oracle's iLQR (the dependency-free C++ restatement of util.cpp + ilqr.py; kind "port": the
    reference's Armadillo/autograd path cannot be built here) on a bounded sample of the same workload, one
    problem per thread on the host cores."""
    from oracle import oracle as O
    from trajopt_amd import envs
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    cores = max(1, min(cores, os.cpu_count() or 1))
    oe = O.make_env(env_name, dt=dt, uw=[uw])
    pe = envs.make(env_name, dt=dt, uw=uw)
    iters = 4
    opts = [1., 1., 1e-6, float("inf"), 1.6, -float("inf"), 0.0, 0.]   # stopping disabled, as the GPU run
    done_steps, elapsed, problems, chunk = 0, 0.0, 0, 4 * cores
    while elapsed < budget_s and problems < 65536:
        x0, kff0 = make_problem_set(pe, chunk, T, seed + 7919 + problems)
        sec, _, it = O.bench_ilqr(oe, T, x0, kff0.reshape(chunk, -1), iters, cores, opts=opts)
        elapsed += sec
        done_steps += it * T
        problems += chunk
    out = {"value": done_steps / elapsed, "unit": "problem-timesteps/s", "cores": cores, "kind": "port",
           "sample": "%d cartpole problems x T=%d x %d iLQR iterations (sequential first-accept line search), "
                     "%.1f s on %d threads, oracle/liboracle.so -O3" % (problems, T, iters, elapsed, cores)}
    # the same port on ONE core, and the NumPy mirror as an indication of the Python-level cost the
    # reference pays per problem (its own autograd path is not runnable here and was not measured)
    x0, kff0 = make_problem_set(pe, 64, T, seed + 104729)
    sec, _, it = O.bench_ilqr(oe, T, x0, kff0.reshape(64, -1), iters, 1, opts=opts)
    out["one_core"] = {"value": it * T / sec, "sample": "64 problems x %d iterations, %.1f s" % (iters, sec)}
    try:
        from oracle import numpy_mirror as Mi
        me = Mi.Env(env_name, dt=dt, uw=[uw])
        Tm = min(T, 50)
        t0 = time.perf_counter()
        Mi.ilqr_run(me, Tm, x0[0], kff0[0][:, :Tm], np.ones(Tm + 1), O.DEFAULT_ALPHAS, 1)
        secm = time.perf_counter() - t0
        out["numpy_mirror"] = {"value": Tm / secm, "sample": "1 problem x T=%d x 1 iteration, %.1f s, 1 core" % (Tm, secm)}
    except Exception as exc:   # the mirror is an extra, never a reason to lose the bench line
        out["numpy_mirror"] = {"value": None, "sample": "failed: %r" % (exc,)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--instances", type=int, default=4096, help="problems per GPU (C2: 4096)")
    ap.add_argument("--horizon", type=int, default=500)
    ap.add_argument("--env", default="cartpole")
    ap.add_argument("--staged", type=int, default=-1, help="line search: -1 auto, 0 one launch, 1 staged, 2 returns first then the accepted alpha again (same results)")
    ap.add_argument("--no-fusion", action="store_true", help="quadratise and backward pass as two launches (same results)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from trajopt_amd import _lib, envs
    from trajopt_amd.ilqr import BatchedILQR

    dt, uw = {"cartpole": (0.02, 1e-5), "pendulum": (0.05, 1e-5), "double_cartpole": (0.01, 1e-5),
              "quadrotor": (0.02, 1e-3)}[a.env]
    env = envs.make(a.env, dt=dt, uw=uw)
    N, T, K, W = a.instances, a.horizon, a.steps, a.warmup
    n, m = env.dm_state, env.dm_act
    x0, kff0 = make_problem_set(env, N, T, 1337 + rank)
    solver = BatchedILQR(env, T, x0, kff_init=kff0, action_penalty=None, apply_env_kwargs=False,
                         tolfun=-float("inf"), tolgrad=0.0, max_lmbda=float("inf"), staged_line_search=a.staged, no_fusion=a.no_fusion,
                         trace_cap=W + K + 16)
    nb_alphas = solver.alphas.size
    L = _lib.lib()
    KERNEL_PHASES = ("quadratize", "backward", "rollout", "select_commit")

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

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    _lib.check(L.trajopt_profile_begin(K))
    _lib.check(L.trajopt_profile_select(C.c_uint(1 << PHASES.index(dom))))
    t0 = time.perf_counter()
    solver.iterate(K)
    fence()
    t1 = time.perf_counter()
    phase_ms = (C.c_double * len(PHASES))()
    nrec = C.c_int(0)
    _lib.check(L.trajopt_profile_end(phase_ms, C.byref(nrec)))
    dom_ms = phase_ms[PHASES.index(dom)] / max(nrec.value, 1)
    KB = min(K, 10)
    phases = profiled(KB)
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    status = solver.host("status")
    iters = solver.host("iters")
    assert np.all(status == 0) and np.all(iters == W + K + KB), "a problem stopped inside the timed region"
    ret = solver.host("last_return")
    assert np.all(np.isfinite(ret)), "non-finite trajectory cost"

    if rank == 0:
        ab = alg_bytes_per_problem_step(n, m, nb_alphas)
        plan = solver.plan()
        mb = moved_bytes_per_problem_step(n, m, plan)
        stamp = _lib.source_stamp()
        is_c2 = a.env == "cartpole" and N == 4096 and T == 500 and nb_alphas == 11
        tag = {("cartpole", 4096, 500): "C2", ("double_cartpole", 8192, 500): "C3 (one GPU's share)",
               ("quadrotor", 4096, 500): "C4", ("pendulum", 1, 100): "C1"}.get((a.env, N, T), "custom")
        # HBM traffic / issue counters of the kernels from the committed rocprofv3 --pmc passes (separate runs, gfx950
        # read correction applied: scripts/pmc_traffic.py, scripts/pmc_sq.py) - PMC counters cannot be sampled from
        # inside this process, so they are quoted only for the workload they were collected on AND only while the
        # kernel sources are the ones they were collected from (stamp)
        pmc = stamped("r02_pmc_traffic_c2.json", stamp) if is_c2 else None
        sq = stamped("r02_pmc_sq_c2.json", stamp) if is_c2 else None
        census = stamped("r02_rollout_isa_census.json", stamp)
        traffic = pmc["by_phase"].get(dom) if pmc else None
        hbm_alg = ab[dom] * N * T / (dom_ms * 1e-3) / 1e9
        if dom == "rollout" and census and a.env in census["kernels"]:
            # the rollout is a chain of plain fp64 vector instructions: priced against the fp64 VALU peak, with the
            # flops of one step counted in the kernel's ISA (profiles/r02_rollout_isa_census.json)
            fl = census["kernels"][a.env]["fp64_flops_per_lane_step"]
            # two-pass line search (plan 2): every (problem, alpha) once for its return, the accepted alpha of each problem
            # once more for its trajectory - the phase holds both launches and the select kernel between them
            lanes = N * nb_alphas + (N if plan["staged_line_search"] == 2 else 0)
            achieved = fl * lanes * T / (dom_ms * 1e-3) / 1e12
            roof = {"bound": "fp64-valu", "kernel": dom, "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / FP64_VALU_PEAK_TFLOPS, "traffic": traffic,
                    "fp64_flops_per_lane_step": fl,
                    # counted in the straight-line form's block loop; the rolled form (oversubscribed batches) runs the
                    # same fp64 arithmetic per step around different loop / addressing code
                    "instructions_per_step": census["kernels"][a.env]["instructions_per_step"] if plan["rollout_straight_line"] else None,
                    "flops_counted_on": "this kernel" if plan["rollout_straight_line"] else "the straight-line form of the same model (same arithmetic per step)",
                    "lanes": lanes, "waves_per_simd": lanes / 64.0 / 1024.0,
                    "launches": "returns of every alpha, select, the accepted alpha again" if plan["staged_line_search"] == 2 else "one",
                    "hbm_algorithmic_GBps": hbm_alg, "hbm_frac": hbm_alg / HBM_PEAK_GBS,
                    "alg_bytes_per_launch": ab[dom] * N * T, "avg_launch_ms": dom_ms,
                    "valu_insts_per_simd_cycle_pmc": (sq["kernels_by_phase"].get(dom, {}).get("valu_insts_per_simd_cycle") if sq else None)}
        else:
            moved = mb.get(dom, ab[dom]) * N * T
            achieved = moved / (dom_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "moved_bytes_per_launch": moved,
                    "alg_bytes_per_launch": ab[dom] * N * T, "avg_launch_ms": dom_ms}
        bwd_moved = mb["backward"] * N * T / (phases["backward"] * 1e-3) / 1e9
        out = {
            "metric": "problem-timesteps/sec (iLQR iteration, batched)",
            "value": world * N * T * K / elapsed,
            "unit": "problem-timesteps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: batched iLQR %s dx=%d du=%d T=%d, %d instances per GPU, %d alphas, reg=1, "
                                   "stopping rules disabled" % (tag, a.env, n, m, T, N, nb_alphas),
                       "instances_per_gpu": N, "horizon": T, "parallelism": "problems sharded x%d, no collective" % world},
            "roofline": roof,
            "phase_ms": phases,
            "phase_ms_note": "six-phase breakdown: %d iterations bracketed after the timed region; the timed region "
                             "carries only the dominant kernel's two events per iteration" % KB,
            "plan": plan,
            "backward_pass": {"moved_bytes_per_problem_step": mb["backward"], "GBps_moved": bwd_moved,
                              "frac_of_hbm_peak": bwd_moved / HBM_PEAK_GBS,
                              "dense_algorithmic_bytes_per_problem_step": ab["backward"],
                              "note": "quadratise + backward in one launch" if plan["fused_quad_bwd"] else "backward kernel alone"},
            "source_stamp": stamp,
            "mean_return": float(ret.mean()),
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.env, dt, uw, T, 1337, a.cpu_budget)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
