// Fused quadratisation + regularised Riccati backward pass for small batches of dx <= 4, du = 1 problems (BASELINE
// config C2: 4096 cartpoles): rows A4 / A5 / A1 of the hot path in ONE launch.
//
//   AnalyticalLinearDynamics / AnalyticalQuadraticCost.taylor_expansion   /root/reference/trajopt/ilqr/objects.py:68-78,114-117
//   backward_pass                                                         /root/reference/trajopt/ilqr/src/util.cpp:123-162
//   lambda retry loop                                                     /root/reference/trajopt/ilqr/ilqr.py:162-173
//
// Why: as two launches the expansion A, B, cx, cu makes a round trip through HBM (0.5 GB written, 0.5 GB read per
// iteration at C2) and the two phases run one after the other although neither fills the chip: the quadratisation is
// HBM-bound, the recursion (bwd_mfma4.hip: four problems per wave on the 4x4x4 matrix cores) is a 500-step latency
// chain of 1024 waves, one per SIMD.  Here every wave alternates between the two jobs for ITS four problems:
//   produce  64 lanes = 16 time steps x 4 problems push the dual numbers of a chunk through clip -> RK4 -> clip and
//            write A_t, B_t, cx_t, cu_t, uref_t, w_t into the wave's own 16 KB of LDS, already in the lane order the
//            matrix-core operands have (an LDS transpose from "lane = (step, problem)" to "lane = (row, problem, col)");
//   consume  the recursion of bwd_mfma4.hip, unchanged in its arithmetic, over those 16 steps, operands read from LDS.
// Nothing is shared between waves, so there is no barrier and no flag anywhere: a wave's LDS writes and reads are
// ordered by its own in-order LDS queue.  What was measured on the way (MI355X, C2): an fp64 instruction occupies a
// SIMD's vector pipe for ~5.3 cycles whichever wave issues it and the recursion alone already keeps it ~75 % busy, so
// a dedicated producer wave per consumer wave (sharing its SIMD, LDS double buffer, one workgroup barrier per chunk)
// overlapped almost nothing - 0.42 ms, the same as the two separate launches - and raising the consumer's priority
// starved the producer (0.61 ms).  One wave per SIMD doing both in turn runs at the pipe's rate: 0.26 ms.
//
// The producer part reads the reference trajectory through the driver's RefView (common.h): from the time-blocked
// candidate plane ref_plane[p] - no copy on accept - or, ref_plane[p] < 0, from the dense xref / uref; what it reads
// from a plane is written through to xref_out / uref_out (first sweep only), where the rollout takes it from.
//
// Used by trajopt_ilqr_iterate for envs with identity cost features (structured cost), PROBLEM_MINOR gains group,
// gains-only outputs; everything else takes the two-kernel path.
#include <type_traits>
#include "sys.h"

namespace trajopt {

#ifndef FUSED_DIAG
#define FUSED_DIAG 0   // timing experiments only: 1 = no expansion arithmetic, 2 = no recursion (wrong results)
#endif
constexpr int kCH = 16;  // time steps per chunk: 16 steps x 4 problems = the 64 lanes of a wave
// doubles per step image: 128 + 1 pad - the expansion's lanes (step, problem) write one element each of 16 different
// step images per instruction, which at a stride of 128 doubles is a 16-way LDS bank conflict
constexpr int kStep = 129;

struct FusedArgs {
    trajopt_env_t env;
    trajopt_backward_io_t io;  // lmbda, dlmbda, dV, K, kff (the candidate gains), diverge, g_norm, active, cost_*
    trajopt_lambda_schedule_t sched;
    const double *xref, *uref;  // reference trajectory of the problems with ref_plane[p] < 0, PROBLEM_MINOR
    RefView rv;                 // rv.ref_plane == nullptr: xref / uref for every problem
    size_t NP;
    int T, retry;
    PostBackward post;          // post.on: the driver's per-problem step behind the backward pass, done here (common.h)
};

__device__ __forceinline__ double f4_mm(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
// Store through a buffer descriptor whose range is the array: a lane that must not store passes an offset beyond it
// and the hardware drops the write.  The recursion's stores are therefore UNCONDITIONAL instructions - under a branch
// hipcc's waitcnt pass cannot count them, and the wait for the prefetched inputs of the next chunk (older than these
// stores in the in-order vmcnt queue) became a wait for every store of the chunk to be acknowledged.
typedef unsigned f4_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f4_rsrc(double* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ void f4_st(double x, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(f4_v2u, x), r, off, 0, 0);
}
constexpr unsigned kNoStore = 0xfffffff8u;  // beyond any array this kernel takes (byte offsets < 2^32 - 8 by the launcher's check)

// LDS image of one time step for one wave: 128 doubles,
//   [lane]       A operand: lane (r, q, c) = 16 r + 4 q + c holds A[r + N c] of problem q
//   [64 + lane]  packed tile P of problem q: row 0 B, row 1 cx, row 2 (cu, uref, w[t], -), row 3 unused
// i.e. exactly the two registers bwd_mfma4.hip loads from HBM per step (structured cost).
template <class SYS, int REG, bool GN>
__global__ __launch_bounds__(256, 1) void fused_qb4_kernel(FusedArgs a) {
    constexpr int N = SYS::N, M = SYS::M;
    static_assert(M == 1 && N >= 2 && N <= 4, "four-problems-per-wave recursion: dx <= 4, du = 1");
    constexpr int KC = N + M;  // all tangent columns of [A | B] in one lane
    __shared__ double tile_all[4][kCH * kStep];

    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    double* tile = &tile_all[wave][0];
    const size_t NP = a.NP;
    const int T = a.T;
    const int nch = (T + kCH - 1) / kCH;
    const trajopt_backward_io_t& io = a.io;
    const trajopt_env_t& e = a.env;
    const unsigned np8 = (unsigned)(NP * 8);
    const size_t En1 = (size_t)N * (T + 1), Em = (size_t)M * T;

    // ---------------- recursion role: lane = 16 r + 4 q + c, q = problem of the wave --------------------------------
    const unsigned r = lane >> 4, q = (lane >> 2) & 3u, c = lane & 3u;
    const size_t cp_raw = (size_t)blockIdx.x * 16 + wave * 4 + q;
    const bool c_exists = cp_raw < NP;
    const size_t cp = c_exists ? cp_raw : NP - 1;  // padding blocks recompute the last problem and store nothing
    const bool in_r = r < (unsigned)N, in_c = c < (unsigned)N, in_rc = in_r && in_c;
    const bool skip = !c_exists || (io.active && io.active[cp] == 0);
    const unsigned cc = in_c ? c : 0u, rr = in_r ? r : 0u, cp8 = (unsigned)(cp * 8);
    // gains: bank io.out_bank[cp] of those interleaved per time step (trajopt_backward_io_t::out_bank)
    const unsigned kmul = io.out_bank ? (unsigned)TRAJOPT_GAIN_BANKS : 1u, bank = c_exists ? (unsigned)gains_bank(io.out_bank, cp) : 0u;
    const unsigned o_c0 = (bank * N + cc) * np8 + cp8, o_z0 = bank * np8 + cp8;  // K[c], kff of slice 0
    const unsigned s_n = kmul * (unsigned)N * np8, s_1 = kmul * np8;
    const bool st_c = (N == 4) || in_c;
    const double eye = (r == c) ? 1.0 : 0.0;
    const double dg = (r == c && in_r) ? io.cost_gxx[r] : 0.0;
    const double guu = io.cost_guu[0];
    const double e0 = r == 0 ? 1.0 : 0.0, e1 = r == 1 ? 1.0 : 0.0, e2 = r == 2 ? 1.0 : 0.0;
    const bool p_live = (r <= 1) ? in_c : (r == 2 ? (c == 0 || (GN && c == 1) || c == 2) : false);
    double lam = io.lmbda[cp];
    double dlam = (a.retry && io.dlmbda) ? io.dlmbda[cp] : 1.0;
    int csel = -1;
    if (a.rv.ref_plane && c_exists) csel = a.rv.ref_plane[cp];
    int div = -1, retries = 0;
    double dV0 = 0.0, dV1 = 0.0, gmax = -INFINITY;
    bool frozen = skip;

    // ---------------- expansion role: lane = 4 s + q: step s of the chunk, problem q ---------------------------------
    const unsigned ps = lane >> 2, pq = lane & 3u;
    const size_t pp_raw = (size_t)blockIdx.x * 16 + wave * 4 + pq;
    const bool p_exists = pp_raw < NP;
    const size_t pp = p_exists ? pp_raw : NP - 1;
    const bool p_livep = p_exists && !(io.active && io.active[pp] == 0);
    int p_sel = -1;
    if (a.rv.ref_plane && p_exists) p_sel = a.rv.ref_plane[pp];
    // the driver: time-blocked candidate plane ref_plane[p] (common.h) when that is >= 0; otherwise the dense xref / uref
    const bool refp = p_sel >= 0;   // per lane
    const double* p_xsrc = refp ? a.rv.cand_state + (size_t)p_sel * tb_plane(N, T + 1, NP) : a.xref;
    const double* p_usrc = refp ? a.rv.cand_action + (size_t)p_sel * tb_plane(M, T, NP) : a.uref;
    const double sc = e.cost_times_dt ? e.dt : 1.0;
    bool first_sweep = true;
    const __amdgpu_buffer_rsrc_t rK = f4_rsrc(io.K, (unsigned)((size_t)kmul * N * M * T * NP * 8));
    const __amdgpu_buffer_rsrc_t rkff = f4_rsrc(io.kff, (unsigned)((size_t)kmul * M * T * NP * 8));
    // uref_out: dropped stores (offset beyond the range) for the lanes with nothing to write through
    const bool through = a.rv.uref_out != nullptr && a.rv.xref_out != nullptr && refp && p_livep;
    const bool have_out = a.rv.uref_out != nullptr && a.rv.xref_out != nullptr;   // (wave-uniform: the descriptors are scalars)
    const __amdgpu_buffer_rsrc_t ruo = f4_rsrc(have_out ? a.rv.uref_out : io.kff, have_out ? (unsigned)((size_t)M * T * NP * 8) : 0u);
    const __amdgpu_buffer_rsrc_t rxo = f4_rsrc(have_out ? a.rv.xref_out : io.kff, have_out ? (unsigned)((size_t)N * (T + 1) * NP * 8) : 0u);

    for (;;) {  // sweeps: one, plus one per lambda retry (ilqr.py:162-173)
        bool alive = !frozen;
        int sdiv = -1;
        double acc0 = 0.0, acc1 = 0.0, gm = 0.0, gnan = 0.0, snap0 = 0.0, snap1 = 0.0;
        double VT, vR;
        unsigned os_c = o_c0 + (unsigned)(T - 1) * s_n, os_z = o_z0 + (unsigned)(T - 1) * s_1;
        {
            // terminal condition (util.cpp:123-124) from the closed-form cost expansion at x_T (objects.py:72-78):
            // V[T] = w[T] diag(gxx), v[T] = cx[T] = w[T] gxx (y_T - g)
            const double wT = io.cost_w[T];
            const double VTerm = wT * dg;
            const double xT = csel >= 0 ? a.rv.cand_state[(size_t)csel * tb_plane(N, T + 1, NP) + tb_index(N, T, (int)rr, cp, NP)]
                                        : a.xref[((size_t)N * T + rr) * NP + cp];
            double vterm = 0.0;
            if (wT != 0.0) {
                const double y = ((e.periodic >> rr) & 1) ? wrap_angle(xT) : xT;
                const double h = wT * io.cost_gxx[rr];
                vterm = h * (y - e.goal[rr]);
            }
            vR = in_r ? vterm : 0.0;
            // x_T is not part of any chunk: written through here
            if (first_sweep && a.rv.xref_out && csel >= 0 && c == 0 && in_r && !skip) a.rv.xref_out[((size_t)N * T + rr) * NP + cp] = xT;
            VT = f4_mm(VTerm, eye, 0.0);
        }

        // `valid` is false for the steps t < 0 of the last, partial chunk (TAIL): they run like the others - the same
        // number of store instructions on every path is what lets the waitcnt pass count them - but store nothing
        // and leave the sums alone.
        auto step = [&](double inA, double inP, int t, auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value;
            const bool valid = !TAIL || t >= 0;
            if (!p_live) inP = 0.0;
            if constexpr (N < 4) {
                if (!in_rc) inA = 0.0;
            }
            const double BR = f4_mm(inP, e0, 0.0), cxR = f4_mm(inP, e1, 0.0), row2 = f4_mm(inP, e2, 0.0);
            const double cuF = f4_mm(e0, row2, 0.0), wF = f4_mm(e2, row2, 0.0);
            const double urF = GN ? f4_mm(e1, row2, 0.0) : 0.0;
            const double ZR = f4_mm(VT, BR, 0.0);
            const double P0 = f4_mm(VT, inA, 0.0);
            const double quF = f4_mm(BR, vR, cuF);
            const double qxR = f4_mm(inA, vR, cxR);
            const double QuuF = f4_mm(BR, ZR, guu);
            const double QxuR = f4_mm(inA, ZR, 0.0);
            const double Qxx = f4_mm(inA, P0, wF * dg);
            const double QxuC = f4_mm(QxuR, eye, 0.0);
            const double Qs = f4_mm(Qxx, eye, Qxx);
            double q_reg = QuuF, QxuR_r = QxuR, QxuC_r = QxuC;
            if constexpr (REG == 2) {
                const double BAr = f4_mm(inA, BR, 0.0);
                QxuR_r = fma(lam, BAr, QxuR);
                QxuC_r = fma(lam, f4_mm(BAr, eye, 0.0), QxuC);
                q_reg = fma(lam, f4_mm(BR, BR, 0.0), QuuF);
            } else if constexpr (REG == 1) {
                q_reg = QuuF + lam;
            }
            const bool bad = __builtin_amdgcn_class(q_reg, 0x27f);  // anything but +denormal | +normal (util.cpp:145)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
                if (alive && bad && valid) { sdiv = t; snap0 = acc0; snap1 = acc1; alive = false; }
            }
            const double rho = rcp_nr(q_reg);
            const double kff = -rho * quF, KC_ = -rho * QxuC_r;
            const double Qk = QuuF * kff;
            if constexpr (TAIL) {
                acc0 = valid ? fma(kff, quF, acc0) : acc0;
                acc1 = valid ? fma(kff, Qk, acc1) : acc1;
            } else {
                acc0 = fma(kff, quF, acc0);  // util.cpp:154; acc1 is halved at the end of the sweep
                acc1 = fma(kff, Qk, acc1);
            }
            if constexpr (REG != 2) {
                const double sig = rho * fma(QuuF, rho, -2.0);
                vR = fma(sig * quF, QxuR, qxR);
                VT = fma(QxuR * QxuC, sig, 0.5 * Qs);
            } else {
                const double KR = -rho * QxuR_r;
                vR = fma(QxuR, kff, fma(KR, Qk + quF, qxR));
                const double cross = KR * QxuC + QxuR * KC_;
                VT = fma(QuuF, KR * KC_, fma(0.5, Qs, cross));
            }
            const bool st = alive && valid;
            f4_st(KC_, rK, (st && st_c) ? os_c : kNoStore);
            f4_st(kff, rkff, st ? os_z : kNoStore);
            if constexpr (GN) {
                const double d = fabs(urF) + 1.0;
                double ri = __builtin_amdgcn_rcp(d);
                ri = fma(ri, fma(-d, ri, 1.0), ri);
                const double g = fabs(kff) * ri;
                gm = st ? fmax(gm, g) : gm;
                gnan = st ? fma(g, 0.0, gnan) : gnan;
            }
            os_c -= s_n;
            os_z -= s_1;
        };

        // Inputs of a chunk's expansion are fetched one chunk ahead (while the previous chunk's recursion runs): issued
        // at the top of a chunk, an HBM round trip per chunk was a quarter of the kernel.  Loads are unconditional, with
        // clamped indices (a load under a branch makes hipcc's waitcnt pass drain the queue at the loop header).
        double nx[N], nu[M], nul[M], nw;
        auto fetch = [&](int jn) {
            int t = T - 1 - kCH * jn - (int)ps;
            t = t < 0 ? 0 : t;
            const int tl = t >= 1 ? t - 1 : 0;
#pragma unroll
            for (int i = 0; i < N; ++i) nx[i] = p_xsrc[refp ? tb_index(N, t, i, pp, NP) : ((size_t)N * t + i) * NP + pp];
#pragma unroll
            for (int k = 0; k < M; ++k) {
                nu[k] = p_usrc[refp ? tb_index(M, t, k, pp, NP) : ((size_t)M * t + k) * NP + pp];
                nul[k] = p_usrc[refp ? tb_index(M, tl, k, pp, NP) : ((size_t)M * tl + k) * NP + pp];
            }
            nw = io.cost_w[t];
        };
        fetch(0);
        // as many (dropped) stores behind the first fetch as a chunk's recursion puts behind every later one: the waitcnt
        // pass merges the loop's two entries, and with no stores on this one it waits for ALL but the newest few
        // operations at the top of a chunk - i.e. for the previous chunk's stores to be acknowledged
#pragma unroll
        for (int k = 0; k < kCH; ++k) { f4_st(0.0, rK, kNoStore); f4_st(0.0, rkff, kNoStore); }
        f4_st(0.0, ruo, kNoStore);
#pragma unroll
        for (int i = 0; i < N; ++i) f4_st(0.0, rxo, kNoStore);
        unsigned long long cyc_prod = 0, cyc_cons = 0, t0_ = 0;
        for (int j = 0; j < nch; ++j) {
            if (FUSED_DIAG == 3) t0_ = __builtin_amdgcn_s_memtime();
            const int tc = T - 1 - kCH * j;  // newest step of the chunk
            // ---- produce: expansion of steps tc .. tc - 15 for the wave's four problems ----
            {
                const int t = tc - (int)ps;
                double x[N], u[M], ul[M];
#pragma unroll
                for (int i = 0; i < N; ++i) x[i] = nx[i];
#pragma unroll
                for (int k = 0; k < M; ++k) { u[k] = nu[k]; ul[k] = (e.slew_rate && t >= 1) ? nul[k] : 0.0; }
                const double w = nw;
                // what was just read, written through for the rollout (unconditional instructions: see f4_st)
                const bool wt = through && first_sweep && t >= 0;
                f4_st(u[0], ruo, wt ? (unsigned)(((size_t)t * NP + pp) * 8) : kNoStore);
#pragma unroll
                for (int i = 0; i < N; ++i) f4_st(x[i], rxo, wt ? (unsigned)((((size_t)N * t + i) * NP + pp) * 8) : kNoStore);
                if (t >= 0 && p_livep) {
                    if (FUSED_DIAG != 1) {
                        double* Lw = tile + ps * kStep;
                        double J[N][KC];
                        SYS::template jacobian_cols<KC>(e, x, u, 0, J);
#pragma unroll
                        for (int k = 0; k < N; ++k)
#pragma unroll
                            for (int i = 0; i < N; ++i) Lw[16 * i + 4 * pq + k] = J[i][k];
#pragma unroll
                        for (int i = 0; i < N; ++i) Lw[64 + 4 * pq + i] = J[i][N];
                        // cost expansion (objects.py:72-78), closed form for identity features
#pragma unroll
                        for (int i = 0; i < N; ++i) {
                            double g = 0.0;
                            if (w != 0.0) {
                                const double y = ((e.periodic >> i) & 1) ? wrap_angle(x[i]) : x[i];
                                const double h = w * (sc * 2.0 * e.gw[i]);
                                g = h * (y - e.goal[i]);
                            }
                            Lw[64 + 16 + 4 * pq + i] = g;
                        }
                        const double d = e.slew_rate ? (u[0] - ul[0]) : u[0];
                        Lw[64 + 32 + 4 * pq + 0] = sc * 2.0 * e.uw[0] * d;
                        Lw[64 + 32 + 4 * pq + 1] = u[0];
                        Lw[64 + 32 + 4 * pq + 2] = w;
                    }
                }
            }
            // the next chunk's inputs: issued here, not before the expansion - there every register is taken and hipcc
            // parks the load destinations in AGPRs at once, i.e. waits for the loads it has just issued
            fetch(j + 1 < nch ? j + 1 : j);
            if (FUSED_DIAG == 3) { __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); cyc_prod += t1_ - t0_; t0_ = t1_; }
            // ---- consume: the recursion over the chunk (the wave's own LDS writes above are ordered before these
            // reads by its in-order LDS queue; the fence keeps the compiler from moving them) ----
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            auto consume = [&](auto tail_c) {
                double inA = tile[lane], inP = tile[64 + lane];
#pragma unroll
                for (int s = 0; s < kCH; ++s) {
                    double nA = 0.0, nP = 0.0;
                    if (s + 1 < kCH) { nA = tile[(s + 1) * kStep + lane]; nP = tile[(s + 1) * kStep + 64 + lane]; }
                    step(inA, inP, tc - s, tail_c);
                    inA = nA; inP = nP;
                }
            };
            if (FUSED_DIAG != 2) {
                if (tc >= kCH - 1) consume(std::false_type{});
                else consume(std::true_type{});
            }
            if (FUSED_DIAG == 3) { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); cyc_cons += t2_ - t0_; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        bool again = false;
        if (FUSED_DIAG == 3) { acc0 = (double)cyc_prod; acc1 = 2.0 * (double)cyc_cons; sdiv = -1; }
        if (!frozen) {
            div = sdiv;
            dV0 = sdiv >= 0 ? snap0 : acc0;
            dV1 = 0.5 * (sdiv >= 0 ? snap1 : acc1);
            gmax = gm + gnan;
            if (a.retry && sdiv >= 0) {  // ilqr.py:166-169
                dlam = fmax(dlam * a.sched.mult_lmbda, a.sched.mult_lmbda);
                lam = fmax(lam * dlam, a.sched.min_lmbda);
                again = (lam <= a.sched.max_lmbda) && (lam < 1e300) && FUSED_DIAG == 0;
            }
        }
        frozen = frozen || !again;
        first_sweep = false;
        // the wave repeats the sweep while any of its blocks may still grow lambda; an exit every wave reaches
        if (__builtin_amdgcn_ballot_w64(again) == 0 || ++retries >= kMaxLambdaRetries) break;
    }
    if (r == 0 && c == 0 && !skip) {
        io.diverge[cp] = div;
        io.dV[cp] = dV0;
        io.dV[NP + cp] = dV1;
        if constexpr (GN) {
            if (io.g_norm) io.g_norm[cp] = FUSED_DIAG == 4 ? (double)retries : gmax;
            // post_backward_kernel's step on what this lane holds (post.on implies retry, dlmbda and g_norm: driver.hip)
            if (a.post.on) post_backward_problem(a.post, a.sched, cp, lam, dlam, gmax, div);
        }
        if (a.retry) {
            io.lmbda[cp] = lam;
            if (io.dlmbda) io.dlmbda[cp] = dlam;
        }
    }
}

template <class SYS>
static constexpr bool fused_qb4_model() {
    return SYS::kStatic && SYS::M == 1 && SYS::N >= 2 && SYS::N <= 4;
}

struct FusedLaunch {
    FusedArgs a;
    int reg;
    bool gn, done;
    hipStream_t st;
    template <class SYS>
    void operator()() {
        if constexpr (fused_qb4_model<SYS>()) {
            const dim3 grid((unsigned)((a.NP + 15) / 16)), block(256);
#define TJ_FUSED(R)                                                                                       \
    if (gn) hipLaunchKernelGGL((fused_qb4_kernel<SYS, R, true>), grid, block, 0, st, a);                  \
    else hipLaunchKernelGGL((fused_qb4_kernel<SYS, R, false>), grid, block, 0, st, a);
            if (reg == 1) { TJ_FUSED(1) }
            else if (reg == 2) { TJ_FUSED(2) }
            else { TJ_FUSED(0) }
#undef TJ_FUSED
            done = true;
        }
    }
};

// Shapes the fused kernel takes: a built-in model with dx <= 4, du = 1 (pendulum, cartpole), batches below the 32 768
// problems from which the lane-per-problem fusion (bwd_tpp.hip) takes over, byte offsets inside one array below 2^32.
// (The four-problems-per-wave recursion ALONE loses to lane = problem beyond ~6k problems - it saturates the texture
// addressers on its dense inputs, see api.hip - but fused it has no such inputs, and what it replaces is two
// launches: 8192 cartpoles 0.97 -> 0.45 ms, 16 384 1.21 -> 0.89 ms, 24 576 1.37 ms; 32 768 lane-per-problem 1.70 ms.)
bool fused_qb4_supported(const trajopt_env_t* env, int T, int64_t NP) {
    if (env->model != TRAJOPT_ENV_PENDULUM && env->model != TRAJOPT_ENV_CARTPOLE) return false;
    if (env->nb_features != 0 || NP >= 32768) return false;
    const int n = env->dm_state;
    return (double)(T + 2) * n * n * (double)NP * 8.0 < 4294967296.0 &&
           (double)TRAJOPT_GAIN_BANKS * (T + 1) * n * (double)NP * 8.0 < 4294967296.0;
}

// io: as for trajopt_ilqr_backward_pass_batched with the structured cost set (cost_w, cost_gxx, cost_guu) and
// gains-only outputs; A, B, cx, cu, Cxx, Cuu, Cxu are not used.
int launch_fused_qb4(const trajopt_env_t* env, const trajopt_backward_io_t& io, int reg, int T, int64_t NP, int retry,
                     const trajopt_lambda_schedule_t& sched, const double* xref, const double* uref,
                     const RefView* rv, hipStream_t st, const PostBackward* post) {
    if (!fused_qb4_supported(env, T, NP) || !io.cost_w) return TRAJOPT_ERR_UNSUPPORTED;
    if (post && post->on && !(retry && io.dlmbda && io.uref && io.g_norm)) return TRAJOPT_ERR_BAD_ARG;
    FusedLaunch L;
    L.a.env = *env; L.a.io = io; L.a.sched = sched; L.a.xref = xref; L.a.uref = uref;
    L.a.rv = rv ? *rv : RefView{nullptr, nullptr, nullptr, nullptr, nullptr};
    L.a.NP = (size_t)NP; L.a.T = T; L.a.retry = retry;
    L.a.post = post ? *post : PostBackward{nullptr, nullptr, nullptr, 0.0, 1, 0, 0};
    L.reg = reg; L.gn = io.uref != nullptr && io.g_norm != nullptr; L.done = false; L.st = st;
    if (!dispatch_model(*env, L) || !L.done) return TRAJOPT_ERR_UNSUPPORTED;
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

}  // namespace trajopt
