// Regularised Riccati backward pass for dx <= 4, du = 1 on the fp64 matrix cores, FOUR problems per wave.
//
// Row A1 of the hot path: the recursion of /root/reference/trajopt/ilqr/src/util.cpp:123-162, with the lambda
// retry loop of /root/reference/trajopt/ilqr/ilqr.py:162-173 optionally folded in.
//
// Why: with lane = problem (bwd_tpp.hip) a step is a chain of ~350 dependent fp64 instructions, so a batch of a
// few thousand problems (C2: 4096 = 64 waves on 1024 SIMDs) is bound by that chain, not by HBM.  Here one
// v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4x4 blocks; block q of a wave is one problem, so a step is
// nine matrix-core instructions plus ~30 vector instructions and the batch spreads over 16x more waves.
// A lone wave on a SIMD pays for EVERY instruction it issues (scalar ones included, ~5-10 cycles each with the
// dependency stalls), so the kernel is written to minimise the instruction count of a step, not its flops.
//
// Register layout (measured, scripts/ubench_mfma4.hip): lane = 16 r + 4 q + c, q = block.  Per block
//     A operand  A_[i][k]  in lane (r = k, c = i)      B operand  B_[k][j]  in lane (r = k, c = j)
//     C / D      D[i][j]   in lane (r = i, c = j)
// i.e. a register that holds X[r][c] is X as a B operand and X^T as an A operand.  With "R" = a vector indexed by
// the lane's r and replicated over c, "C" = indexed by c and replicated over r, "F" = one scalar in all 16 lanes,
// and E = the identity:
//     P0   = V A              mfma(V,  A,  0)        V symmetric, so V^T = V
//     ZR   = V B              mfma(V,  BR, 0)        BR[r][c] = B[r]
//     Qxx  = Cxx + A'P0       mfma(A,  P0, Cxx)      Qs = Qxx + Qxx' = mfma(Qxx, E, Qxx)
//     QxuR = Cxu + A'Z        mfma(A,  ZR, CxuR)     QxuC = mfma(QxuR, E, 0)      (the same numbers, column-indexed)
//     QuuF = Cuu + B'Z        mfma(BR, ZR, CuuF)     quF  = cu + B'v = mfma(BR, vR, cuF)
//     qxR  = cx + A'v         mfma(A,  vR, cxR)
// so every quantity the rank-one update V = Qxx + K'QuuK + K'Qux + Qux'K needs at lane (r, c) - K[r], K[c], Qxu[r],
// Qxu[c], Quu - comes out of the matrix cores already in the lane that uses it: no cross-lane moves at all, and
// one load per input array and step.
//
// Data: PROBLEM_MINOR (element e of step t for 16 consecutive problems = one 128-byte line).  A workgroup is four
// waves = 16 consecutive problems, so the four waves together consume whole lines through their CU's L1.
// A step is only a few hundred cycles, far less than the HBM latency, so the inputs are prefetched kRing steps
// ahead into a ring of register sets (the time loop is unrolled kRing times; 8 loads x kRing < 64 = the vmcnt range).
// Addresses are SGPR base + one 32-bit VGPR byte offset per element pattern, stepped back one slice per step.
// Divergence (util.cpp:145-148) is per block: the wave keeps sweeping in lockstep, a failed block only stops
// storing; the lambda retry repeats the sweep for the wave while any of its blocks still has to grow lambda.
#include "smallmat.h"

namespace trajopt {

struct M4Args {
    trajopt_backward_io_t io;
    trajopt_lambda_schedule_t sched;
    size_t NP;
    int T, reg, retry;
};

__device__ __forceinline__ double m4_mm(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
// uniform base + 32-bit per-lane byte offset: global_load/store with an SGPR base, no 64-bit vector arithmetic
__device__ __forceinline__ double m4_ld(const double* base, unsigned off) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ void m4_st(double x, double* base, unsigned off) {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = x;
}

// One step's inputs.  A (and, for a dense cost, Cxx) are loaded in the lane layout of the operand they feed.  All the
// vectors and scalars of the step travel in ONE packed register P, a 4 x 4 tile per problem,
//     row 0: B[0..3]    row 1: cx[0..3]    row 2: cu, uref, (w[t] | Cuu), -    row 3: Cxu[0..3] (dense cost only)
// and the matrix cores replicate them into the layouts the step needs (with e_k the register "1 where r == k"):
//     mfma(P, e_k, 0)[i][j] = P[k][i]   (row k as an "R" vector)      mfma(e_k, X, 0)[i][j] = X[k][j]
// because the texture addressers take a quarter-wave per cycle whatever the addresses: seven loads per step cost every
// wave 112 cycles of the CU's one address pipeline, two cost 32, and the six extra matrix instructions are off the
// critical path.
struct M4In {
    double A, P, Cxx;
};
constexpr int kRing = 7;  // prefetch distance in steps

// Byte offsets of one time slice for the four element patterns (r + N c, r, c, 0); stepped by the slice strides.
struct M4Off {
    unsigned rc, r, c, z;
};

// SC: structured cost (trajopt_backward_io_t::cost_w) - Cxx = w[t] diag(gxx), Cuu = guu, Cxu = 0 are not loaded.
// `pp` is the lane's own address inside P's source array for the slice being loaded.
template <int N, bool SC>
__device__ __forceinline__ void m4_load(M4In& s, const trajopt_backward_io_t& io, const M4Off& o, const char* pp,
                                        bool in_rc, bool p_live) {
    // loads only: the padding lanes (which read a valid address) are zeroed where the values are used - a select
    // here would make the wave wait for the load it has just issued
    s.A = m4_ld(io.A, o.rc);
    s.P = *reinterpret_cast<const double*>(pp);
    if constexpr (!SC) s.Cxx = m4_ld(io.Cxx, o.rc);
}

// REG: 1 / 2 = the reference's regularisation modes (util.cpp:135-143), 0 = none (any other value of `reg`).
// __launch_bounds__(256, 2): at most 256 VGPRs, which lets hipcc keep the MFMA results in VGPRs instead of AGPRs
// (no v_accvgpr_read/write around every matrix-core instruction).
template <int N, int REG, bool FULL, bool GN, bool SC>
__global__ __launch_bounds__(256, 2) void bwd_mfma4_kernel(M4Args a) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned r = lane >> 4, q = (lane >> 2) & 3u, c = lane & 3u;
    const size_t NP = a.NP;
    const size_t p_raw = (size_t)blockIdx.x * 16 + wave * 4 + q;
    const bool exists = p_raw < NP;
    const size_t p = exists ? p_raw : NP - 1;  // padding blocks recompute the last problem and store nothing
    const int T = a.T;
    const bool in_r = r < (unsigned)N, in_c = c < (unsigned)N, in_rc = in_r && in_c;
    const bool skip = !exists || (a.io.active && a.io.active[p] == 0);
    const trajopt_backward_io_t& io = a.io;

    const unsigned rr = in_r ? r : 0u, cc = in_c ? c : 0u;  // padding lanes read element 0 (then zeroed)
    const unsigned np8 = (unsigned)(NP * 8), p8 = (unsigned)(p * 8);
    const M4Off o0{(rr + N * cc) * np8 + p8, rr * np8 + p8, cc * np8 + p8, p8};  // slice t = 0
    const unsigned s_nn = (unsigned)(N * N) * np8, s_n = (unsigned)N * np8, s_1 = np8;
    auto slice = [&](int t) {
        const unsigned tn = (unsigned)t;
        return M4Off{o0.rc + tn * s_nn, o0.r + tn * s_n, o0.c + tn * s_n, o0.z + tn * s_1};
    };
    auto back = [&](M4Off& o) { o.rc -= s_nn; o.r -= s_n; o.c -= s_n; o.z -= s_1; };
    // gains: bank io.out_bank[p] of those interleaved per time step (trajopt_backward_io_t::out_bank); du = 1
    const unsigned kmul = a.io.out_bank ? (unsigned)TRAJOPT_GAIN_BANKS : 1u, bank = (unsigned)gains_bank(a.io.out_bank, p);
    const unsigned gK0 = (bank * N + cc) * np8 + p8, gk0 = bank * np8 + p8, gs_n = kmul * s_n, gs_1 = kmul * s_1;
    // Stores need no lane predicate for N = 4: a value replicated over r (or c, or the whole block) is written by
    // all its holders to the one address - same value, same address, merged by the memory pipeline.
    const bool st_rc = (N == 4) || in_rc, st_r = (N == 4) || in_r, st_c = (N == 4) || in_c;

    double lam = io.lmbda[p];
    double dlam = (a.retry && io.dlmbda) ? io.dlmbda[p] : 1.0;
    int div = -1, retries = 0;
    double dV0 = 0.0, dV1 = 0.0, gmax = -INFINITY;
    bool frozen = skip;  // frozen blocks take part in the wave's arithmetic but keep their results and store nothing
    const double eye = (r == c) ? 1.0 : 0.0;  // identity as a B operand: mfma(X, eye, 0) = X^T
    // structured cost: Cxx[t] = w[t] dg with dg = diag(gxx) in the C-operand layout, Cuu = guu
    const double dg = (SC && r == c && in_r) ? io.cost_gxx[r] : 0.0;
    const double guu = SC ? io.cost_guu[0] : 0.0;
    const double e0 = r == 0 ? 1.0 : 0.0, e1 = r == 1 ? 1.0 : 0.0, e2 = r == 2 ? 1.0 : 0.0, e3 = r == 3 ? 1.0 : 0.0;
    // the lane's slot of the packed tile P: source array, byte offset inside slice 0, bytes per slice
    const char* p_base;
    size_t p_off, p_stride;
    bool p_live = true;
    {
        const size_t vec_off = ((size_t)cc * NP + p) * 8, vec_stride = (size_t)N * NP * 8;  // element c of an n-vector
        const size_t sc_off = p * 8, sc_stride = NP * 8;                                    // a per-problem scalar
        p_base = reinterpret_cast<const char*>(io.B); p_off = vec_off; p_stride = vec_stride;
        if (r == 0) {
            p_live = in_c;
        } else if (r == 1) {
            p_base = reinterpret_cast<const char*>(io.cx); p_live = in_c;
        } else if (r == 2) {
            p_base = reinterpret_cast<const char*>(io.cu); p_off = sc_off; p_stride = sc_stride;
            if (c == 1) {
                if (GN) p_base = reinterpret_cast<const char*>(io.uref);
                p_live = GN;
            } else if (c == 2) {
                if (SC) { p_base = reinterpret_cast<const char*>(io.cost_w); p_off = 0; p_stride = 8; }
                else p_base = reinterpret_cast<const char*>(io.Cuu);
            } else if (c == 3) {
                p_live = false;
            }
        } else {
            if (!SC) p_base = reinterpret_cast<const char*>(io.Cxu);
            p_live = !SC && in_c;
        }
    }

    for (;;) {
        bool alive = !frozen;
        int sdiv = -1;                                        // this sweep's failing step
        double acc0 = 0.0, acc1 = 0.0, gm = 0.0, gnan = 0.0;  // this sweep's dV and gradient-norm accumulators
        double snap0 = 0.0, snap1 = 0.0;                      // their values when the block failed
        // terminal condition (util.cpp:123-124): V[T] = Cxx[T] as given; the A-operand image of V is its transpose
        double VT, vR;
        {
            const M4Off oT = slice(T);
            double VTerm = SC ? m4_ld(io.cost_w, (unsigned)T * 8u) * dg : m4_ld(io.Cxx, oT.rc);
            vR = m4_ld(io.cx, oT.r);
            if constexpr (N < 4) {
                if (!in_rc) VTerm = 0.0;
                if (!in_r) vR = 0.0;
            }
            if constexpr (FULL) {
                if (alive) {
                    if (st_rc && io.V) m4_st(VTerm, io.V, oT.rc);
                    if (st_r && io.v) m4_st(vR, io.v, oT.r);
                }
            }
            VT = m4_mm(VTerm, eye, 0.0);
        }
        M4Off ol = slice(T - 1), os = ol;  // slice being loaded / slice being computed and stored
        unsigned gK = gK0 + (unsigned)(T - 1) * gs_n, gk = gk0 + (unsigned)(T - 1) * gs_1;  // K[c], kff of that slice
        const char* pl = p_base + p_off + (size_t)(T - 1) * p_stride;  // the lane's slot of P in the slice being loaded
        M4In ring[kRing];
#pragma unroll
        for (int k = 0; k < kRing; ++k) {
            // unconditional (a load under a branch makes the waitcnt pass drain the queue at the loop header): a
            // horizon shorter than the ring re-loads slice 0 into the slots it never uses
            m4_load<N, SC>(ring[k], io, ol, pl, in_rc, p_live);
            if (T - 2 - k >= 0) { back(ol); pl -= p_stride; }
            // keep the prologue's loads in slot order, the order the loop refills them in: the waitcnt pass merges
            // both at the loop header and otherwise drains every load (vmcnt(0)) once per trip round the ring
            __builtin_amdgcn_sched_barrier(0);
        }

        auto step = [&](const M4In& raw, int t) {
            // unpack P (independent of V: the scheduler is free to run these ahead of the chain)
            M4In in = raw;
            if (!p_live) in.P = 0.0;  // unused slots of the packed tile
            if constexpr (N < 4) {    // padding lanes of a smaller state: zero rows / columns
                if (!in_rc) { in.A = 0.0; if constexpr (!SC) in.Cxx = 0.0; }
            }
            const double BR = m4_mm(in.P, e0, 0.0), cxR = m4_mm(in.P, e1, 0.0), row2 = m4_mm(in.P, e2, 0.0);
            const double cuF = m4_mm(e0, row2, 0.0), x22F = m4_mm(e2, row2, 0.0);  // x22 = w[t] (SC) or Cuu
            const double urF = GN ? m4_mm(e1, row2, 0.0) : 0.0;
            const double CxuR = SC ? 0.0 : m4_mm(in.P, e3, 0.0);
            const double ZR = m4_mm(VT, BR, 0.0);
            const double P0 = m4_mm(VT, in.A, 0.0);
            const double quF = m4_mm(BR, vR, cuF);
            const double qxR = m4_mm(in.A, vR, cxR);
            const double QuuF = m4_mm(BR, ZR, SC ? guu : x22F);
            const double QxuR = m4_mm(in.A, ZR, CxuR);
            const double Qxx = m4_mm(in.A, P0, SC ? x22F * dg : in.Cxx);
            const double QxuC = m4_mm(QxuR, eye, 0.0);  // D[i][j] = Qxu[j]
            const double Qs = m4_mm(Qxx, eye, Qxx);     // Qxx + Qxx' (one rounding per entry: bitwise symmetric)
            // regularisation (util.cpp:135-143)
            double q_reg = QuuF, QxuR_r = QxuR, QxuC_r = QxuC;
            if constexpr (REG == 2) {  // (V + lam I): Qux_reg = Qux + lam B'A, Quu_reg = Quu + lam B'B
                const double BAr = m4_mm(in.A, BR, 0.0);
                QxuR_r = fma(lam, BAr, QxuR);
                QxuC_r = fma(lam, m4_mm(BAr, eye, 0.0), QxuC);
                q_reg = fma(lam, m4_mm(BR, BR, 0.0), QuuF);
            } else if constexpr (REG == 1) {
                q_reg = QuuF + lam;
            }
            if constexpr (FULL) {  // Q-function blocks are stored before the PD test, like the reference
                if (alive) {
                    if (st_rc && io.Qxx) m4_st(Qxx, io.Qxx, os.rc);
                    if (st_c && io.Qux) m4_st(QxuC, io.Qux, os.c);
                    if (st_r && io.qx) m4_st(qxR, io.qx, os.r);
                    if (io.Quu) m4_st(QuuF, io.Quu, os.z);
                    if (io.qu) m4_st(quF, io.qu, os.z);
                }
            }
            // PD test: Armadillo's is_sympd for a 1 x 1 matrix (util.cpp:145), q - 100 eps |q| > 0, holds exactly for
            // the positive finite q (+inf gives inf - inf).  A failure is rare: one wave-uniform branch per step,
            // and the failing block's accumulators are snapshotted there instead of masked every step.
            const bool bad = __builtin_amdgcn_class(q_reg, 0x27f);  // anything but +denormal | +normal
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
                if (alive && bad) { sdiv = t; snap0 = acc0; snap1 = acc1; alive = false; }
            }
            // gains (util.cpp:150-152) and the value update (util.cpp:154-161)
            const double rho = rcp_nr(q_reg);
            const double kff = -rho * quF, KC = -rho * QxuC_r;
            const double Qk = QuuF * kff;
            acc0 = fma(kff, quF, acc0);  // util.cpp:154; acc1 is halved at the end of the sweep
            acc1 = fma(kff, Qk, acc1);
            if constexpr (REG != 2) {
                // K = -rho Qxu, kff = -rho qu  =>  K'Quu K + K'Qux + Qux'K = sig Qxu Qxu',  v = qx + sig qu Qxu
                // with sig = rho (rho Quu - 2); Qxu[r] Qxu[c] is a commutative product, so V stays bitwise symmetric
                // (V^T = V is what lets V serve as its own A-operand image)
                const double sig = rho * fma(QuuF, rho, -2.0);
                vR = fma(sig * quF, QxuR, qxR);
                VT = fma(QxuR * QxuC, sig, 0.5 * Qs);
            } else {
                const double KR = -rho * QxuR_r;
                vR = fma(QxuR, kff, fma(KR, Qk + quF, qxR));             // util.cpp:156-157
                const double cross = KR * QxuC + QxuR * KC;               // commutative sum: symmetric
                VT = fma(QuuF, KR * KC, fma(0.5, Qs, cross));             // util.cpp:159-161
            }
            if (alive) {
                if (st_c) m4_st(KC, io.K, gK);
                m4_st(kff, io.kff, gk);
                if constexpr (FULL) {
                    if (st_rc && io.V) m4_st(VT, io.V, os.rc);
                    if (st_r && io.v) m4_st(vR, io.v, os.r);
                }
                if constexpr (GN) {  // ilqr.py:176; NaN-propagating max: gnan turns NaN once any ratio is NaN
                    const double d = fabs(urF) + 1.0;
                    double ri = __builtin_amdgcn_rcp(d);
                    ri = fma(ri, fma(-d, ri, 1.0), ri);
                    const double g = fabs(kff) * ri;
                    gm = fmax(gm, g);
                    gnan = fma(g, 0.0, gnan);
                }
            }
            back(os);
            gK -= gs_n;
            gk -= gs_1;
        };

        // Slot k of the ring holds step t; once consumed it is refilled IN PLACE with step t - kRing (a refill issued
        // before the slot's last use needs fresh registers, and the copies back at the loop latch wait for every
        // outstanding load).  Whole rounds whose refills all exist run in a loop with ONE exit, at its latch (any
        // other exit edge is routed through the latch by the compiler and makes the waitcnt pass assume slot 0 may
        // have been refilled last, draining the queue every round); the last 7..13 steps are straight-line code.
        int t = T - 1;
        for (int round = (T >= kRing ? (T - kRing) / kRing : 0); round > 0; --round) {
#pragma unroll
            for (int k = 0; k < kRing; ++k) {
                step(ring[k], t);
                m4_load<N, SC>(ring[k], io, ol, pl, in_rc, p_live);
                back(ol);
                pl -= p_stride;
                --t;
            }
        }
#pragma unroll
        for (int k = 0; k < 2 * kRing - 1; ++k) {
            if (t >= 0) {
                step(ring[k % kRing], t);
                if (k < kRing - 1 && t - kRing >= 0) {
                    m4_load<N, SC>(ring[k % kRing], io, ol, pl, in_rc, p_live);
                    back(ol);
                    pl -= p_stride;
                }
                --t;
            }
        }

        bool again = false;
        if (!frozen) {
            div = sdiv;
            dV0 = sdiv >= 0 ? snap0 : acc0;
            dV1 = 0.5 * (sdiv >= 0 ? snap1 : acc1);
            gmax = gm + gnan;  // gnan is 0 or NaN
            // ilqr.py:166-169 for the blocks that failed
            if (a.retry && sdiv >= 0) {
                dlam = fmax(dlam * a.sched.mult_lmbda, a.sched.mult_lmbda);
                lam = fmax(lam * dlam, a.sched.min_lmbda);
                again = (lam <= a.sched.max_lmbda) && (lam < 1e300);
            }
        }
        frozen = frozen || !again;
        // the wave repeats the sweep while any of its blocks may still grow lambda; an exit every wave reaches:
        // retries are bounded even if lambda can grow for ever (max_lmbda = inf)
        if (__builtin_amdgcn_ballot_w64(again) == 0 || ++retries >= kMaxLambdaRetries) break;
    }
    if (r == 0 && c == 0 && !skip) {
        io.diverge[p] = div;
        io.dV[p] = dV0;
        io.dV[NP + p] = dV1;
        if (a.retry) {
            io.lmbda[p] = lam;
            if (io.dlmbda) io.dlmbda[p] = dlam;
        }
        if constexpr (GN) {
            if (io.g_norm) io.g_norm[p] = gmax;  // mean over du = 1 actions of max_t |kff| / (|uref| + 1)  (ilqr.py:176)
        }
    }
}

template <int N, int REG, bool SC>
static int m4_launch_nrs(const M4Args& a, hipStream_t st) {
    const bool full = a.io.Qxx || a.io.Qux || a.io.Quu || a.io.qx || a.io.qu || a.io.V || a.io.v;
    const bool gn = a.io.uref != nullptr && a.io.g_norm != nullptr;
    const dim3 block(256), grid((unsigned)((a.NP + 15) / 16));
    if (full) {
        if (gn) hipLaunchKernelGGL((bwd_mfma4_kernel<N, REG, true, true, SC>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bwd_mfma4_kernel<N, REG, true, false, SC>), grid, block, 0, st, a);
    } else {
        if (gn) hipLaunchKernelGGL((bwd_mfma4_kernel<N, REG, false, true, SC>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bwd_mfma4_kernel<N, REG, false, false, SC>), grid, block, 0, st, a);
    }
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}
template <int N, int REG>
static int m4_launch_nr(const M4Args& a, hipStream_t st) {
    return a.io.cost_w ? m4_launch_nrs<N, REG, true>(a, st) : m4_launch_nrs<N, REG, false>(a, st);
}
template <int N>
static int m4_launch_n(const M4Args& a, hipStream_t st) {
    if (a.reg == 1) return m4_launch_nr<N, 1>(a, st);
    if (a.reg == 2) return m4_launch_nr<N, 2>(a, st);
    return m4_launch_nr<N, 0>(a, st);
}

// PROBLEM_MINOR only (the caller checks), du = 1, dx <= 4, and every byte offset inside one array below 2^32.
int launch_backward_mfma4(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int retry,
                          const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled) {
    *handled = false;
    if (m != 1 || n < 2 || n > 4) return TRAJOPT_OK;
    if ((double)(T + 2) * n * n * (double)NP * 8.0 >= 4294967296.0) return TRAJOPT_OK;
    if (io.out_bank && (double)TRAJOPT_GAIN_BANKS * (T + 1) * n * (double)NP * 8.0 >= 4294967296.0) return TRAJOPT_OK;
    M4Args a;
    a.io = io; a.sched = sched; a.NP = (size_t)NP; a.T = T; a.reg = reg; a.retry = retry;
    *handled = true;
    if (n == 2) return m4_launch_n<2>(a, st);
    if (n == 3) return m4_launch_n<3>(a, st);
    return m4_launch_n<4>(a, st);
}

}  // namespace trajopt
