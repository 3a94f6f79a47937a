// Batched iLQR driver (row A6): the per-problem control flow of iLQR.run,
// /root/reference/trajopt/ilqr/ilqr.py:136-237, kept entirely on the device.
//
// An iteration is four or five launches on one stream and no host synchronisation:
//   quadratise -> backward pass (lambda retries inside the kernel, ilqr.py:162-173)   [one fused launch for small
//        dx <= 4 batches] -> gradient-norm stop (ilqr.py:176-180) -> rollout of every alpha at once (ilqr.py:187-191)
//        -> first-accept selection + lambda update + stopping rules (ilqr.py:193-231).
// Accepting a candidate (ilqr.py:210-217) copies nothing: iteration i writes candidates and gains into bank i % 2 and
// an accept only sets ref_plane[p] / gains_bank[p] (include/trajopt_amd.h, ABI v5); a problem that has rejected every
// alpha keeps its reference in xref / uref (written through by the quadratisation) and writes its next gains into a
// third bank; xref / K / kff are brought up to date once per trajopt_ilqr_iterate call.  Problems that have stopped keep their slot and are masked out through `active`
// (SURVEY 9.5).  Everything is PROBLEM_MINOR except the gains group (ws->gains_layout, see include/trajopt_amd.h).
#include "sys.h"

namespace trajopt {

int launch_rollout(const trajopt_env_t* env, int T, int64_t NP, int layout, const double* x0, const double* K,
                   const double* kff, const double* xref, const double* uref, const double* weighting,
                   const double* alphas_host, int nb_alphas, const int32_t* active, double* state, double* action,
                   double* cost, double* ret, int32_t* finite, hipStream_t st, int k_layout, int alpha_begin,
                   int alpha_end, const int32_t* plist, const int32_t* pcount, const RolloutPlanes* planes);
int launch_quadratize(const trajopt_env_t* env, int T, int64_t NP, int layout, int convention, const double* xref,
                      const double* uref, const double* weighting, const int32_t* active, double* A, double* B,
                      double* c, double* Cxx, double* cx, double* Cuu, double* cu, double* Cxu, hipStream_t st,
                      const RefView* rv, int out_layout);

bool fused_qb4_supported(const trajopt_env_t* env, int T, int64_t NP);
bool fused_tpp_supported(const trajopt_env_t* env, int T, int64_t NP);
int launch_fused_tpp(const trajopt_env_t* env, const trajopt_backward_io_t& io, int reg, int T, int64_t NP, int retry,
                     const trajopt_lambda_schedule_t& sched, const double* xref, const double* uref, const RefView* rv,
                     hipStream_t st);
bool rollout_straight_line(int n, int m, size_t NP, int nb_alphas);
bool rollout_cost_only_supported(const trajopt_env_t* env, int T, int64_t NP, int k_layout);
int backward_kernel_family(int n, int m, int T, int64_t NP, int layout, int want, bool structured);
int launch_fused_qb4(const trajopt_env_t* env, const trajopt_backward_io_t& io, int reg, int T, int64_t NP, int retry,
                     const trajopt_lambda_schedule_t& sched, const double* xref, const double* uref,
                     const RefView* rv, hipStream_t st, const PostBackward* post);

struct DrvArgs {
    trajopt_ilqr_options_t opt;
    trajopt_ilqr_workspace_t ws;
    int n, m;
};

// (grow_lambda / shrink_lambda, post_backward_problem: common.h)

// dV = [kff'qu, 1/2 kff'Quu kff] (util.cpp:154) of problem p as the backward kernels store it: dV belongs to the gains
// group, so it is (2, NP) struct-of-arrays for PROBLEM_MINOR and (NP, 2) for PROBLEM_MAJOR.
__device__ __forceinline__ void load_dV(const DrvArgs& a, size_t p, size_t NP, double& dV0, double& dV1) {
    const bool gminor = a.ws.gains_layout == TRAJOPT_LAYOUT_PROBLEM_MINOR;
    dV0 = a.ws.dV[gminor ? p : 2 * p];
    dV1 = a.ws.dV[gminor ? NP + p : 2 * p + 1];
}

// ilqr.py:139-149: first alpha whose whole state trajectory is < 1e8 becomes the reference.  The return must also be
// finite: all(state < 1e8) lets -inf / hugely negative states through, whose cost the device sincos reports as NaN.
__global__ void init_select_kernel(DrvArgs a) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t NP = (size_t)a.ws.nb_problems;
    if (p >= NP) return;
    int sel = -1;
    for (int k = 0; k < a.opt.nb_alphas; ++k)
        if (a.ws.cand_finite[(size_t)k * NP + p] && isfinite(a.ws.cand_ret[(size_t)k * NP + p])) { sel = k; break; }
    a.ws.sel[p] = sel;
    // the initial rollout wrote its candidates into bank 0: plane `sel` there is published to xref / uref right away
    // (none finite: they keep what they hold), trajopt_ilqr_init_rollout then resets ref_plane to -1
    a.ws.ref_plane[p] = sel;
    a.ws.gains_bank[p] = -1;
    // no finite alpha: last_return keeps its previous value (-inf on a fresh solver, ilqr.py:87)
    const double r = sel >= 0 ? a.ws.cand_ret[(size_t)sel * NP + p] : a.ws.last_return[p];
    a.ws.last_return[p] = r;
    a.ws.trace[p] = r;  // ilqr.py:149
    a.ws.trace_len[p] = 1;
    a.ws.status[p] = 0;
    a.ws.iters[p] = 0;
}

// Start of an iteration (ilqr.py:151): a problem takes part while its status is 0.
// wb: the bank the coming iteration writes.  A problem's gains go there too - unless that bank holds its accepted
// controller (it has rejected every alpha since): then to the spare bank, so that nothing accepted is ever overwritten.
__device__ __forceinline__ void prep_problem(const DrvArgs& a, size_t p, int wb) {
    const int act = a.ws.status[p] == 0 ? 1 : 0;
    a.ws.active[p] = act;
    if (act) a.ws.iters[p] += 1;
    a.ws.out_bank[p] = a.ws.gains_bank[p] == wb ? TRAJOPT_GAIN_BANKS - 1 : wb;
}
__global__ void prep_kernel(DrvArgs a, int wb) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)a.ws.nb_problems) return;
    prep_problem(a, p, wb);
}

// ilqr.py:175-180 plus the "backward pass gave up" path (:168 -> :224-231).
// Also, for bank wb, which this iteration's rollout is about to write: a reference that still lives in one of its
// planes (the problem rejected every alpha since) is from now on the copy the quadratisation has just written through to
// xref / uref.
__global__ void post_backward_kernel(DrvArgs a, int wb) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)a.ws.nb_problems) return;
    if (!a.ws.active[p]) return;
    const PostBackward pb{a.ws.ref_plane, a.ws.status, a.ws.active, a.opt.tolgrad, a.opt.nb_alphas, wb, 1};
    double lam = a.ws.lmbda[p], dlam = a.ws.dlmbda[p];
    post_backward_problem(pb, a.opt.sched, p, lam, dlam, a.ws.g_norm[p], a.ws.diverge[p]);
    a.ws.lmbda[p] = lam;
    a.ws.dlmbda[p] = dlam;
}

// Staged line search, stage 1 -> 2: problems whose first alpha is rejected (ilqr.py:196-202) are appended to a
// compact list; only they roll out the remaining alphas.  List order depends on the atomics' arrival order, the
// per-problem results do not.
__global__ void pending_kernel(DrvArgs a) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t NP = (size_t)a.ws.nb_problems;
    if (p >= NP) return;
    if (!a.ws.active[p]) return;
    const double alpha = a.opt.alphas[0];
    const double dret = a.ws.last_return[p] - a.ws.cand_ret[p];
    double dV0, dV1;
    load_dV(a, p, NP, dV0, dV1);
    const double expected = -1.0 * alpha * (dV0 + alpha * dV1);
    const bool accepted = dret / expected >= a.opt.min_imp;
    // the decision is RECORDED (sel: 0 = alphas[0] accepted, -1 = pending) and select_kernel takes it from there
    // instead of evaluating the test a second time: nothing forces two compilations of it to round alike, and a
    // problem left off the list but rejected by select would go on to read the stale returns of the other alphas
    a.ws.sel[p] = accepted ? 0 : -1;
    if (!accepted) {
        const int j = atomicAdd(a.ws.pending_count, 1);
        a.ws.pending_list[j] = (int32_t)p;
    }
}

// ilqr.py:187-231: sequential first-accept over the alphas, evaluated from the precomputed returns.
// prep_next: another iteration follows in this call - its prep (same thread-per-problem mapping, nothing reads `active`
// in between) is done here instead of in a launch of its own.
// ls: line_search_mode() - 1 (staged): pending_kernel has already decided alphas[0] for this problem (sel[p]); 2 (two
// passes): the accepted trajectory is about to be written to plane 0 of the bank, whatever its alpha.
// wb: the bank this iteration wrote.
__device__ __forceinline__ void select_problem(const DrvArgs& a, size_t p, size_t NP, int ls, int wb);
__global__ void select_kernel(DrvArgs a, int prep_next, int ls, int wb) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t NP = (size_t)a.ws.nb_problems;
    if (p >= NP) return;
    select_problem(a, p, NP, ls, wb);
    if (prep_next) prep_problem(a, p, wb ^ 1);
}
__device__ __forceinline__ void select_problem(const DrvArgs& a, size_t p, size_t NP, int ls, int wb) {
    const bool staged = ls == 1;
    const bool first_accepted = staged && a.ws.sel[p] == 0;
    a.ws.sel[p] = -1;
    if (!a.ws.active[p]) return;
    const double last = a.ws.last_return[p];
    double dV0, dV1;
    load_dV(a, p, NP, dV0, dV1);
    int sel = -1;
    double best = 0.0, dret_sel = 0.0, alpha_last = a.ws.alpha[p];
    // the first 16 returns are fetched up front: behind the loop's early exit they were one dependent L2 round trip
    // per step size (this kernel is nothing but that latency)
    constexpr int kPre = 16;
    double pre[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) pre[k] = k < a.opt.nb_alphas ? a.ws.cand_ret[(size_t)k * NP + p] : 0.0;
    for (int k = 0; k < a.opt.nb_alphas; ++k) {
        const double alpha = a.opt.alphas[k];
        double ret;
        if (k < kPre) {
            ret = pre[0];
#pragma unroll
            for (int j = 1; j < kPre; ++j) ret = (k == j) ? pre[j] : ret;
        } else {
            ret = a.ws.cand_ret[(size_t)k * NP + p];
        }
        const double dret = last - ret;
        const double expected = -1.0 * alpha * (dV0 + alpha * dV1);
        const double imp = dret / expected;
        if (sel < 0) alpha_last = alpha;  // self.alpha follows the loop variable (ilqr.py:188)
        const bool ok = (staged && k == 0) ? first_accepted : (imp >= a.opt.min_imp);  // NaN compares false -> rejected
        if (ok) {
            if (!a.opt.argmin_alpha) { sel = k; dret_sel = dret; break; }
            if (sel < 0 || ret < best) { sel = k; best = ret; dret_sel = dret; alpha_last = alpha; }
        }
    }
    double lam = a.ws.lmbda[p], dlam = a.ws.dlmbda[p];
    if (sel >= 0) {
        shrink_lambda(lam, dlam, a.opt.sched);
        const double ret = a.ws.cand_ret[(size_t)sel * NP + p];
        a.ws.last_return[p] = ret;
        const int len = a.ws.trace_len[p];
        if (len < a.ws.trace_cap) {
            a.ws.trace[(size_t)len * NP + p] = ret;  // ilqr.py:219
            a.ws.trace_len[p] = len + 1;
        }
        if (dret_sel < a.opt.tolfun) a.ws.status[p] = 2;  // ilqr.py:222
        a.ws.sel[p] = sel;
        // ilqr.py:210-217 without a copy: the candidate's plane IS the reference from now on and the bank the backward
        // pass wrote this problem's gains to holds the accepted controller
        a.ws.ref_plane[p] = wb * a.opt.nb_alphas + (ls == 2 ? 0 : sel);
        a.ws.gains_bank[p] = a.ws.out_bank[p];
    } else {
        grow_lambda(lam, dlam, a.opt.sched);
        if (lam > a.opt.sched.max_lmbda) a.ws.status[p] = 3;  // ilqr.py:228
    }
    a.ws.alpha[p] = alpha_last;
    a.ws.lmbda[p] = lam;
    a.ws.dlmbda[p] = dlam;
}

// Index of element e of step t of problem p in a gains-group array (eps elements per step, `slices` steps).
template <bool GMINOR>
__device__ __forceinline__ size_t gidx(size_t eps, size_t slices, int t, size_t e, size_t p, size_t NP) {
    return GMINOR ? (eps * (size_t)t + e) * NP + p : p * eps * slices + eps * (size_t)t + e;
}
// The same inside bank b of the multi-buffered cand_K / cand_kff (trajopt_backward_io_t::out_bank).
template <bool GMINOR>
__device__ __forceinline__ size_t gidx_bank(size_t eps, size_t slices, int t, size_t e, size_t p, size_t NP, int b) {
    return GMINOR ? (eps * (size_t)(TRAJOPT_GAIN_BANKS * t + b) + e) * NP + p
                  : ((size_t)b * NP + p) * eps * slices + eps * (size_t)t + e;
}

// Bring xref / uref (and, with_ctl, K / kff) up to date with the accepted iterate (ilqr.py:210-217 / :142-143): copy
// plane ref_plane[p] and the gains of bank gains_bank[p].
// Thread per (problem, step); once per driver call, not per iteration.
template <bool GMINOR>
__global__ void publish_kernel(DrvArgs a, int with_ctl) {
    const size_t NP = (size_t)a.ws.nb_problems;
    const int T = a.ws.nb_steps, n = a.n, m = a.m;
    // grid = (problem blocks, T + 1): no 64-bit division in the index math
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = (int)blockIdx.y;
    if (p >= NP) return;
    const int ref = a.ws.ref_plane[p];
    if (ref >= 0) {
        const double* st = a.ws.cand_state + (size_t)ref * tb_plane(n, T + 1, NP);
        for (int i = 0; i < n; ++i) a.ws.xref[((size_t)n * t + i) * NP + p] = st[tb_index(n, t, i, p, NP)];
        if (t < T) {
            const double* ac = a.ws.cand_action + (size_t)ref * tb_plane(m, T, NP);
            for (int j = 0; j < m; ++j) a.ws.uref[((size_t)m * t + j) * NP + p] = ac[tb_index(m, t, j, p, NP)];
        }
    }
    const int b = with_ctl ? a.ws.gains_bank[p] : -1;
    if (b < 0 || t >= T) return;
    const size_t nm = (size_t)n * m;
    for (size_t k = 0; k < nm; ++k) a.ws.K[gidx<GMINOR>(nm, T, t, k, p, NP)] = a.ws.cand_K[gidx_bank<GMINOR>(nm, T, t, k, p, NP, b)];
    for (size_t k = 0; k < (size_t)m; ++k) a.ws.kff[gidx<GMINOR>(m, T, t, k, p, NP)] = a.ws.cand_kff[gidx_bank<GMINOR>(m, T, t, k, p, NP, b)];
}
// The same for PROBLEM_MAJOR gains: K (n m T doubles) and kff (m T) of a problem are one contiguous run each, copied with
// consecutive threads on consecutive elements.  (publish_kernel's thread per (problem, step) puts every 8-byte access of a
// wave into a different cache line: 2.8 ms for 4096 quadrotors x T = 500, 0.6 TB/s - a third of a millisecond per iteration
// of an 8-iteration call.)  grid = (problems, pieces of a run).
__global__ __launch_bounds__(256) void publish_gains_major_kernel(DrvArgs a) {
    const size_t NP = (size_t)a.ws.nb_problems, p = (size_t)blockIdx.x;
    if (p >= NP) return;
    const int b = a.ws.gains_bank[p];
    if (b < 0) return;
    const size_t T = (size_t)a.ws.nb_steps, nK = (size_t)a.n * a.m * T, nk = (size_t)a.m * T;
    {
        const double* src = a.ws.cand_K + ((size_t)b * NP + p) * nK;
        double* dst = a.ws.K + p * nK;
        const size_t per = (nK + gridDim.y - 1) / gridDim.y, lo = (size_t)blockIdx.y * per, hi = lo + per < nK ? lo + per : nK;
        for (size_t e = lo + threadIdx.x; e < hi; e += blockDim.x) dst[e] = src[e];
    }
    {
        const double* src = a.ws.cand_kff + ((size_t)b * NP + p) * nk;
        double* dst = a.ws.kff + p * nk;
        const size_t per = (nk + gridDim.y - 1) / gridDim.y, lo = (size_t)blockIdx.y * per, hi = lo + per < nk ? lo + per : nk;
        for (size_t e = lo + threadIdx.x; e < hi; e += blockDim.x) dst[e] = src[e];
    }
}
// ... after which every reference and every accepted controller is in the dense arrays
__global__ void published_kernel(DrvArgs a) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)a.ws.nb_problems) return;
    a.ws.ref_plane[p] = -1;
    a.ws.gains_bank[p] = -1;
}

// keep_value only: promote the value / Q-function arrays of the problems that accepted in this iteration
// (ilqr.py:214-215).  Thread per (problem, step).
template <bool GMINOR>
__global__ void promote_value_kernel(DrvArgs a) {
    const size_t NP = (size_t)a.ws.nb_problems;
    const int T = a.ws.nb_steps, n = a.n, m = a.m;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = (int)blockIdx.y;
    if (p >= NP) return;
    if (a.ws.sel[p] < 0) return;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
#define TJ_COPY(dst, src, eps, slices, cnt) \
    for (size_t k = 0; k < (cnt); ++k) { const size_t o = gidx<GMINOR>(eps, slices, t, k, p, NP); a.ws.dst[o] = a.ws.src[o]; }
    TJ_COPY(V, cand_V, nn, T + 1, nn)
    TJ_COPY(v, cand_v, n, T + 1, (size_t)n)
    if (t < T) {
        TJ_COPY(Qxx, cand_Qxx, nn, T, nn)
        TJ_COPY(Qux, cand_Qux, nm, T, nm)
        TJ_COPY(Quu, cand_Quu, mm, T, mm)
        TJ_COPY(qx, cand_qx, n, T, (size_t)n)
        TJ_COPY(qu, cand_qu, m, T, (size_t)m)
    }
#undef TJ_COPY
}

// ilqr.py:176 for the PROBLEM_MAJOR gains group (the backward kernels only fuse it when kff and uref share a
// layout): g_norm[p] = mean_j max_t |kff_jt| / (|uref_jt| + 1).  One wave per problem.
__global__ __launch_bounds__(64) void gnorm_major_kernel(DrvArgs a) {
    const size_t NP = (size_t)a.ws.nb_problems, p = blockIdx.x;
    if (!a.ws.active[p]) return;
    const int T = a.ws.nb_steps, m = a.m, lane = threadIdx.x;
    double acc = 0.0;
    for (int j = 0; j < m; ++j) {
        double mx = -INFINITY;
        bool nan_seen = false;
        for (int t = lane; t < T; t += 64) {
            const double q = fabs(a.ws.cand_kff[((size_t)a.ws.out_bank[p] * NP + p) * (size_t)m * T + (size_t)m * t + j]) /
                             (fabs(a.ws.uref[((size_t)m * t + j) * NP + p]) + 1.0);
            nan_seen = nan_seen || (q != q);
            mx = fmax(mx, q);
        }
        for (int o = 32; o >= 1; o >>= 1) {
            mx = fmax(mx, __shfl_xor(mx, o, 64));
            nan_seen = nan_seen || __shfl_xor((int)nan_seen, o, 64);
        }
        acc += nan_seen ? NAN : mx;
    }
    if (lane == 0) a.ws.g_norm[p] = acc / m;
}

// ---- optional per-phase timing of trajopt_ilqr_iterate (bench.py's roofline leg) ------------------------
// hipEvents recorded on the launch stream around each phase of an iteration.  Events are created by
// trajopt_profile_begin(), never inside iterate; with profiling off iterate records nothing.
constexpr int kPhases = 6;  // prep | quadratise | backward | post-backward | rollout | select+commit
struct Profile {
    bool on = false;
    int cap = 0, used = 0;
    hipEvent_t* ev = nullptr;  // (kPhases + 1) events per recorded iteration
    unsigned mask = (1u << kPhases) - 1;  // phases that are bracketed (trajopt_profile_select)
};
static Profile g_prof;
// boundary k separates phase k-1 from phase k: recorded when either neighbour is selected.  An event costs the stream
// about 5 us on this GPU (7 per iteration: 0.926 vs 0.888 ms on the bench workload), hence the selection.
static inline void prof_mark(hipStream_t st, int slot, int k) {
    if (slot < 0) return;
    const bool need = (k < kPhases && ((g_prof.mask >> k) & 1u)) || (k > 0 && ((g_prof.mask >> (k - 1)) & 1u));
    if (need) (void)hipEventRecord(g_prof.ev[(size_t)slot * (kPhases + 1) + k], st);
}

static int check_ws(const trajopt_env_t* env, const trajopt_ilqr_options_t* opt, const trajopt_ilqr_workspace_t* ws) {
    if (!env || !opt || !ws) return TRAJOPT_ERR_BAD_ARG;
    if (ws->nb_steps <= 0 || ws->nb_problems <= 0 || opt->nb_alphas <= 0 || opt->nb_alphas > TRAJOPT_MAX_ALPHAS)
        return TRAJOPT_ERR_BAD_ARG;
    const void* need[] = {ws->x0, ws->weighting, ws->xref, ws->uref, ws->K, ws->kff, ws->last_return, ws->lmbda,
                          ws->dlmbda, ws->alpha, ws->status, ws->iters, ws->trace_len, ws->trace, ws->A, ws->B,
                          ws->Cxx, ws->cx, ws->Cuu, ws->cu, ws->Cxu, ws->cand_K, ws->cand_kff, ws->dV, ws->g_norm,
                          ws->diverge, ws->active, ws->sel, ws->cand_state, ws->cand_action, ws->cand_ret,
                          ws->cand_finite, ws->ref_plane, ws->gains_bank, ws->out_bank};
    for (const void* q : need)
        if (!q) return TRAJOPT_ERR_BAD_ARG;
    if (opt->keep_value) {
        const void* need2[] = {ws->V, ws->v, ws->Qxx, ws->Qux, ws->Quu, ws->qx, ws->qu, ws->cand_V, ws->cand_v,
                               ws->cand_Qxx, ws->cand_Qux, ws->cand_Quu, ws->cand_qx, ws->cand_qu};
        for (const void* q : need2)
            if (!q) return TRAJOPT_ERR_BAD_ARG;
    }
    if (ws->trace_cap < 1) return TRAJOPT_ERR_BAD_ARG;
    if (ws->gains_layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && ws->gains_layout != TRAJOPT_LAYOUT_PROBLEM_MINOR)
        return TRAJOPT_ERR_BAD_ARG;
    return TRAJOPT_OK;
}

}  // namespace trajopt

using namespace trajopt;

extern "C" {

int trajopt_ilqr_init_rollout(const trajopt_env_t* env, const trajopt_ilqr_options_t* opt,
                              const trajopt_ilqr_workspace_t* ws, void* stream) {
    int rc = check_ws(env, opt, ws);
    if (rc != TRAJOPT_OK) return rc;
    hipStream_t st = as_stream(stream);
    const int64_t NP = ws->nb_problems;
    const int T = ws->nb_steps, n = env->dm_state, m = env->dm_act;
    // candidates go, time-blocked, to bank 0; the reference is the dense xref / uref
    const RolloutPlanes rp{0, 2 * opt->nb_alphas, nullptr};
    rc = launch_rollout(env, T, NP, TRAJOPT_LAYOUT_PROBLEM_MINOR, ws->x0, ws->K, ws->kff, ws->xref, ws->uref,
                        ws->weighting, opt->alphas, opt->nb_alphas, nullptr, ws->cand_state, ws->cand_action, nullptr,
                        ws->cand_ret, ws->cand_finite, st, ws->gains_layout, 0, opt->nb_alphas, nullptr, nullptr, &rp);
    if (rc != TRAJOPT_OK) return rc;
    DrvArgs a{*opt, *ws, n, m};
    const unsigned g1 = (unsigned)((NP + 255) / 256);
    const dim3 g2(g1, (unsigned)(T + 1));
    hipLaunchKernelGGL(init_select_kernel, dim3(g1), dim3(256), 0, st, a);
    hipLaunchKernelGGL(publish_kernel<true>, g2, dim3(256), 0, st, a, 0);   // xref / uref = the accepted rollout
    hipLaunchKernelGGL(published_kernel, dim3(g1), dim3(256), 0, st, a);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

// How an iteration evaluates its step sizes (trajopt_ilqr_options_t::staged_line_search; reported by trajopt_ilqr_plan):
//   0  one launch: every (problem, alpha) rolls out and writes its candidate trajectory;
//   1  staged: alphas[0] for everybody, the remaining ones only for the problems that rejected it - worth its second
//      500-step latency chain, and the scattered accesses of its compacted second stage, only once the one-launch form
//      oversubscribes the chip (1024 SIMDs x 64 lanes) many times over - measured on 65 536 cartpoles x 11 alphas early in
//      an optimisation (most problems reject alphas[0]): one launch 8.6 ms, staged 12.6 ms; never with argmin (needs
//      every alpha) or a single alpha;
//   2  two passes: the returns of every (problem, alpha) without writing a trajectory, then the accepted step size of
//      each problem once more, written.  The stores of the nb_alphas - 1 candidates nobody reads are most of what a
//      large rollout moves (65 536 cartpoles: 14.4 GB of 18.3 per launch); the price is a second latency chain of one
//      lane per problem, so it is taken from 1.4 waves per SIMD on (small models; 6 for the others).
static int line_search_mode(const trajopt_env_t* env, const trajopt_ilqr_options_t* opt, const trajopt_ilqr_workspace_t* ws) {
    const int64_t lanes = ws->nb_problems * (int64_t)opt->nb_alphas;
    const bool can_stage = !opt->argmin_alpha && opt->nb_alphas >= 2 && ws->pending_list && ws->pending_count;
    const bool can_two = opt->nb_alphas >= 2 && rollout_cost_only_supported(env, ws->nb_steps, ws->nb_problems, ws->gains_layout);
    if (opt->staged_line_search == 1) return can_stage ? 1 : 0;
    if (opt->staged_line_search == 2) return can_two ? 2 : 0;
    if (opt->staged_line_search == 0) return 0;
    // (measured, 11 alphas, T = 500, ms an iteration: cartpoles 8192: 2.21 -> 2.08, 10 240: 2.51 -> 2.18, 16 384: 3.34 ->
    // 2.76 - while 4096, whose waves have a SIMD each, would pay 0.4 ms for the second chain; double cartpoles, whose
    // second chain is 0.9 ms, lose at 8192 - 3.52 -> 4.07 - and at 16 384 - 5.90 -> 6.21 - and win 5 % at 32 768)
    if (can_two && lanes >= (env->dm_state + env->dm_act <= 5 ? 11 * 8192 : 6 * 65536)) return 2;
    // (131 072 cartpoles: staged 21.4 ms an iteration, two passes 16.2)
    if (can_stage && lanes >= 16 * 65536) return 1;
    return 0;
}

int trajopt_ilqr_plan(const trajopt_env_t* env, const trajopt_ilqr_options_t* opt, const trajopt_ilqr_workspace_t* ws,
                      trajopt_ilqr_plan_t* plan) {
    if (!plan) return TRAJOPT_ERR_BAD_ARG;
    const int rc = check_ws(env, opt, ws);
    if (rc != TRAJOPT_OK) return rc;
    const int n = env->dm_state, m = env->dm_act, T = ws->nb_steps, GL = ws->gains_layout;
    const int64_t NP = ws->nb_problems;
    const bool gminor = GL == TRAJOPT_LAYOUT_PROBLEM_MINOR;
    plan->structured_cost = env->nb_features == 0 && trajopt_backward_supports_structured_cost(n, m, T, NP, GL) != 0;
    const bool may_fuse = plan->structured_cost && gminor && !opt->keep_value && !opt->no_fusion;
    plan->fused_quad_bwd = may_fuse && fused_qb4_supported(env, T, NP) ? 1 : (may_fuse && fused_tpp_supported(env, T, NP) ? 2 : 0);
    const int ls = line_search_mode(env, opt, ws);
    const bool staged = ls == 1;
    plan->staged_line_search = ls;
    plan->backward_kernel = plan->fused_quad_bwd == 1 ? TRAJOPT_KERNEL_MFMA4 : plan->fused_quad_bwd == 2 ? TRAJOPT_KERNEL_LANE
                                                      : backward_kernel_family(n, m, T, NP, GL, TRAJOPT_KERNEL_AUTO, plan->structured_cost != 0);
    // (the fast rollout variant at all: built-in model, plain cost - rollout.hip)
    const bool fast = env->model != TRAJOPT_ENV_LQR && !env->slew_rate && env->periodic == 0 && !env->cost_times_dt && env->nb_features == 0;
    plan->rollout_straight_line = fast && ls == 0 && rollout_straight_line(n, m, (size_t)NP, opt->nb_alphas);
    plan->pad_ = 0;
    return TRAJOPT_OK;
}

int trajopt_ilqr_iterate(const trajopt_env_t* env, const trajopt_ilqr_options_t* opt,
                         const trajopt_ilqr_workspace_t* ws, int nb_iter, void* stream) {
    int rc = check_ws(env, opt, ws);
    if (rc != TRAJOPT_OK) return rc;
    hipStream_t st = as_stream(stream);
    const int64_t NP = ws->nb_problems;
    const int T = ws->nb_steps, n = env->dm_state, m = env->dm_act;
    DrvArgs a{*opt, *ws, n, m};
    const unsigned g1 = (unsigned)((NP + 255) / 256);
    const dim3 g2(g1, (unsigned)(T + 1));
    trajopt_backward_io_t io{};
    io.Cxx = ws->Cxx; io.cx = ws->cx; io.Cuu = ws->Cuu; io.cu = ws->cu; io.Cxu = ws->Cxu; io.A = ws->A; io.B = ws->B;
    io.lmbda = ws->lmbda; io.dlmbda = ws->dlmbda;
    if (opt->keep_value) {
        io.Qxx = ws->cand_Qxx; io.Qux = ws->cand_Qux; io.Quu = ws->cand_Quu; io.qx = ws->cand_qx; io.qu = ws->cand_qu;
        io.V = ws->cand_V; io.v = ws->cand_v;
    }
    io.dV = ws->dV; io.K = ws->cand_K; io.kff = ws->cand_kff; io.diverge = ws->diverge;
    const int GL = ws->gains_layout;
    const bool gminor = GL == TRAJOPT_LAYOUT_PROBLEM_MINOR;
    io.g_norm = ws->g_norm; io.active = ws->active;
    io.uref = gminor ? ws->uref : nullptr;  // fused gradient norm needs kff and uref in one layout
    // Structured cost: with identity features (not the ...WithCartesianCost envs) Cxx = w[t] 2 diag(gw), Cuu = 2 diag(uw), Cxu = 0
    // (quadratize.hip); when the backward kernel of this shape can form them itself they are neither written by the
    // quadratisation nor read back - 168 of the 368 bytes a cartpole step moves through HBM each way.
    const bool structured = env->nb_features == 0 && trajopt_backward_supports_structured_cost(n, m, T, NP, GL) != 0;
    if (structured) {
        const double sc = env->cost_times_dt ? env->dt : 1.0;
        io.cost_w = ws->weighting;
        for (int i = 0; i < n; ++i) io.cost_gxx[i] = sc * 2.0 * env->gw[i];
        for (int j = 0; j < m; ++j) io.cost_guu[j] = sc * 2.0 * env->uw[j];
    }
    const bool may_fuse = structured && gminor && !opt->keep_value && !opt->no_fusion;
    const bool fused_qb = may_fuse && fused_qb4_supported(env, T, NP);
    const bool fused_lane = may_fuse && !fused_qb && fused_tpp_supported(env, T, NP);
    const int ls = line_search_mode(env, opt, ws);
    const bool staged = ls == 1, two_pass = ls == 2;
    // the four-problems-per-wave fused kernel also does the per-problem step behind the backward pass (it needs the
    // gradient norm it computes itself: uref and g_norm given)
    const bool post_fused = fused_qb && io.uref != nullptr && io.g_norm != nullptr;
    // No copy on accept: the quadratisation reads the reference through ref_plane (time-blocked planes: whole 64-byte
    // runs per problem, whatever planes its neighbours read) and writes what it read through to the dense xref / uref,
    // where the backward kernels' gradient norm and the rollout find it; iteration `it` of this call writes its
    // candidates into bank it % 2 and its gains into bank out_bank[p] (it % 2 for all but the rejecting problems).
    const RefView rv{ws->ref_plane, ws->cand_state, ws->cand_action, ws->xref, ws->uref};
    io.out_bank = ws->out_bank;
    // xref / uref / K / kff of the accepted iterate (and both maps back to -1, the state of a fresh workspace)
    auto publish = [&]() {
        if (gminor) hipLaunchKernelGGL(publish_kernel<true>, g2, dim3(256), 0, st, a, 1);
        else if (NP <= 0x7fffffff) {   // xref / uref by the thread-per-(problem, step) kernel, the gains as contiguous runs
            hipLaunchKernelGGL(publish_kernel<false>, g2, dim3(256), 0, st, a, 0);
            hipLaunchKernelGGL(publish_gains_major_kernel, dim3((unsigned)NP, 8), dim3(256), 0, st, a);
        } else hipLaunchKernelGGL(publish_kernel<false>, g2, dim3(256), 0, st, a, 1);
        hipLaunchKernelGGL(published_kernel, dim3(g1), dim3(256), 0, st, a);
    };
    // An error inside the loop still publishes what has been accepted so far: the maps then point at nothing
    // a later call could overwrite, and xref / uref / K / kff are those of the last accepted iterate.  The caller
    // should nevertheless treat the workspace as needing a reset (BatchedILQR marks itself uninitialised).
    auto fail = [&](int code) { publish(); (void)hipGetLastError(); return code; };
    for (int it = 0; it < nb_iter; ++it) {
        const int slot = (g_prof.on && g_prof.used < g_prof.cap) ? g_prof.used++ : -1;
        const int wb = it & 1;
        const RolloutPlanes rp{wb * opt->nb_alphas, 2 * opt->nb_alphas, ws->out_bank};
        prof_mark(st, slot, 0);
        if (it == 0) hipLaunchKernelGGL(prep_kernel, dim3(g1), dim3(256), 0, st, a, wb);   // later ones ride in select_kernel
        prof_mark(st, slot, 1);
        if (fused_qb) {
            // small batches of dx <= 4, du = 1 models: quadratisation and recursion in one launch, the expansion
            // handed over through LDS (fused_qb4.hip); reported under the "backward" phase
            prof_mark(st, slot, 2);
            // (... and what post_backward_kernel does, in the lane that writes a problem's scalars: one launch less)
            const PostBackward pb{ws->ref_plane, ws->status, ws->active, opt->tolgrad, opt->nb_alphas, wb, post_fused ? 1 : 0};
            rc = launch_fused_qb4(env, io, opt->reg, T, NP, 1, opt->sched, ws->xref, ws->uref, &rv, st, &pb);
            if (rc != TRAJOPT_OK) return fail(rc);
        } else if (fused_lane) {
            // large batches of the same models: lane = problem, the expansion in registers (bwd_tpp.hip)
            prof_mark(st, slot, 2);
            rc = launch_fused_tpp(env, io, opt->reg, T, NP, 1, opt->sched, ws->xref, ws->uref, &rv, st);
            if (rc != TRAJOPT_OK) return fail(rc);
        } else {
            rc = launch_quadratize(env, T, NP, TRAJOPT_LAYOUT_PROBLEM_MINOR, 0, ws->xref, ws->uref, ws->weighting,
                                   ws->active, ws->A, ws->B, nullptr, structured ? nullptr : ws->Cxx, ws->cx,
                                   structured ? nullptr : ws->Cuu, ws->cu, structured ? nullptr : ws->Cxu, st, &rv, GL);
            if (rc != TRAJOPT_OK) return fail(rc);
            prof_mark(st, slot, 2);
            rc = trajopt_ilqr_backward_pass_batched(&io, opt->reg, n, m, T, NP, GL, 1, &opt->sched, st);
            if (rc != TRAJOPT_OK) return fail(rc);
        }
        if (!gminor) hipLaunchKernelGGL(gnorm_major_kernel, dim3((unsigned)NP), dim3(64), 0, st, a);
        prof_mark(st, slot, 3);
        if (!post_fused) hipLaunchKernelGGL(post_backward_kernel, dim3(g1), dim3(256), 0, st, a, wb);
        prof_mark(st, slot, 4);
        if (two_pass) {
            // pass 1: the returns of every (problem, alpha), nothing else written
            const RolloutPlanes rp1{rp.first_plane, rp.junk_plane, rp.out_bank, 1, nullptr};
            rc = launch_rollout(env, T, NP, TRAJOPT_LAYOUT_PROBLEM_MINOR, ws->x0, ws->cand_K, ws->cand_kff, ws->xref,
                                ws->uref, ws->weighting, opt->alphas, opt->nb_alphas, ws->active, ws->cand_state,
                                ws->cand_action, nullptr, ws->cand_ret, ws->cand_finite, st, GL, 0, opt->nb_alphas,
                                nullptr, nullptr, &rp1);
            if (rc != TRAJOPT_OK) return fail(rc);
        } else if (!staged) {
            rc = launch_rollout(env, T, NP, TRAJOPT_LAYOUT_PROBLEM_MINOR, ws->x0, ws->cand_K, ws->cand_kff, ws->xref,
                                ws->uref, ws->weighting, opt->alphas, opt->nb_alphas, ws->active, ws->cand_state,
                                ws->cand_action, nullptr, ws->cand_ret, ws->cand_finite, st, GL, 0, opt->nb_alphas,
                                nullptr, nullptr, &rp);
            if (rc != TRAJOPT_OK) return fail(rc);
        } else {
            rc = launch_rollout(env, T, NP, TRAJOPT_LAYOUT_PROBLEM_MINOR, ws->x0, ws->cand_K, ws->cand_kff, ws->xref,
                                ws->uref, ws->weighting, opt->alphas, opt->nb_alphas, ws->active, ws->cand_state,
                                ws->cand_action, nullptr, ws->cand_ret, ws->cand_finite, st, GL, 0, 1, nullptr, nullptr,
                                &rp);
            if (rc != TRAJOPT_OK) return fail(rc);
            if (const hipError_t e = hipMemsetAsync(ws->pending_count, 0, sizeof(int32_t), st); e != hipSuccess) {
                set_hip_error(e, "hipMemsetAsync(pending_count)");
                return fail(TRAJOPT_ERR_HIP);
            }
            hipLaunchKernelGGL(pending_kernel, dim3(g1), dim3(256), 0, st, a);
            rc = launch_rollout(env, T, NP, TRAJOPT_LAYOUT_PROBLEM_MINOR, ws->x0, ws->cand_K, ws->cand_kff, ws->xref,
                                ws->uref, ws->weighting, opt->alphas, opt->nb_alphas, ws->active, ws->cand_state,
                                ws->cand_action, nullptr, ws->cand_ret, ws->cand_finite, st, GL, 1, opt->nb_alphas,
                                ws->pending_list, ws->pending_count, &rp);
            if (rc != TRAJOPT_OK) return fail(rc);
        }
        if (!two_pass) prof_mark(st, slot, 5);
        hipLaunchKernelGGL(select_kernel, dim3(g1), dim3(256), 0, st, a, (it + 1 < nb_iter && !two_pass) ? 1 : 0, ls, wb);
        if (two_pass) {
            // pass 2: the accepted step size of every problem once more (same arithmetic, same results), this time
            // written - to plane 0 of the bank, which select_kernel has just made the reference (ONE plane for every
            // alpha: neighbouring problems stay neighbours in memory); it reads the gains through
            // out_bank, so the next iteration's prep (which moves out_bank on) follows it instead of riding in select.
            // Reported under the "rollout" phase.
            const RolloutPlanes rp2{rp.first_plane, rp.junk_plane, rp.out_bank, 0, ws->sel};
            rc = launch_rollout(env, T, NP, TRAJOPT_LAYOUT_PROBLEM_MINOR, ws->x0, ws->cand_K, ws->cand_kff, ws->xref,
                                ws->uref, ws->weighting, opt->alphas, opt->nb_alphas, ws->active, ws->cand_state,
                                ws->cand_action, nullptr, nullptr, nullptr, st, GL, 0, opt->nb_alphas, nullptr, nullptr,
                                &rp2);
            if (rc != TRAJOPT_OK) return fail(rc);
            prof_mark(st, slot, 5);
            if (it + 1 < nb_iter) hipLaunchKernelGGL(prep_kernel, dim3(g1), dim3(256), 0, st, a, wb ^ 1);
        }
        if (opt->keep_value) {
            if (gminor) hipLaunchKernelGGL(promote_value_kernel<true>, g2, dim3(256), 0, st, a);
            else hipLaunchKernelGGL(promote_value_kernel<false>, g2, dim3(256), 0, st, a);
        }
        if (it == nb_iter - 1) publish();   // once per call
        prof_mark(st, slot, 6);
        // a failed launch is reported in the iteration it happens in, not after the remaining ones ran on its garbage
        if (const hipError_t e = hipGetLastError(); e != hipSuccess) {
            set_hip_error(e, "launch inside trajopt_ilqr_iterate");
            return fail(TRAJOPT_ERR_HIP);
        }
    }
    return TRAJOPT_OK;
}

int trajopt_profile_begin(int max_iterations) {
    if (max_iterations <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (g_prof.ev) {
        for (int i = 0; i < g_prof.cap * (kPhases + 1); ++i) (void)hipEventDestroy(g_prof.ev[i]);
        delete[] g_prof.ev;
        g_prof.ev = nullptr;
    }
    g_prof.ev = new hipEvent_t[(size_t)max_iterations * (kPhases + 1)];
    for (int i = 0; i < max_iterations * (kPhases + 1); ++i) TRAJOPT_HIP_CHECK(hipEventCreate(&g_prof.ev[i]));
    g_prof.cap = max_iterations;
    g_prof.used = 0;
    g_prof.mask = (1u << kPhases) - 1;
    g_prof.on = true;
    return TRAJOPT_OK;
}

int trajopt_profile_select(unsigned phase_mask) {
    if (!g_prof.ev || phase_mask == 0 || phase_mask >= (1u << kPhases)) return TRAJOPT_ERR_BAD_ARG;
    g_prof.mask = phase_mask;
    return TRAJOPT_OK;
}

int trajopt_profile_end(double* phase_ms, int* nb_iterations) {
    if (!phase_ms || !nb_iterations || !g_prof.ev) return TRAJOPT_ERR_BAD_ARG;
    g_prof.on = false;
    for (int k = 0; k < kPhases; ++k) phase_ms[k] = 0.0;
    for (int i = 0; i < g_prof.used; ++i) {
        hipEvent_t* e = g_prof.ev + (size_t)i * (kPhases + 1);
        for (int k = 0; k < kPhases; ++k) {
            if (!((g_prof.mask >> k) & 1u)) continue;
            TRAJOPT_HIP_CHECK(hipEventSynchronize(e[k + 1]));
            float ms = 0.f;
            TRAJOPT_HIP_CHECK(hipEventElapsedTime(&ms, e[k], e[k + 1]));
            phase_ms[k] += (double)ms;
        }
    }
    *nb_iterations = g_prof.used;
    return TRAJOPT_OK;
}

}  // extern "C"
