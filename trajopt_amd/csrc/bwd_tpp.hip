// Thread-per-problem regularised Riccati backward pass for tiny state dimensions (dx <= 6).
//
// Row A1 of the hot path: the recursion of /root/reference/trajopt/ilqr/src/util.cpp:123-162, with the
// lambda retry loop of /root/reference/trajopt/ilqr/ilqr.py:162-173 optionally folded in.
//
// Mapping: lane = problem, PROBLEM_MINOR layout, so element e of step t for 64 consecutive problems
// is one contiguous 512-byte row: every global access of a wave is fully coalesced.  V, v and all
// per-step temporaries live in VGPRs (all loops unroll).  The time loop is unrolled by two with two
// named input register sets: the loads of step t-1 are issued before the arithmetic of step t, so
// HBM latency hides under the ~250 dependent fp64 FMAs of a step.
#include <cstdlib>
#include <type_traits>
#include "smallmat.h"
#include "sys.h"

namespace trajopt {

template <int N, int M>
struct StepIn {
    double A[N * N], B[N * M], cx[N], cu[M], ur[M];
};
// the cost blocks of a step: read once, as the starting values of Qxx / Quu / Qxu, and dead from there on
template <int N, int M>
struct CostIn {
    double Cxx[N * N], Cuu[M * M], Cxu[N * M];
};

struct TppArgs {
    trajopt_backward_io_t io;
    trajopt_lambda_schedule_t sched;
    size_t NP;
    int T, reg, retry;
    int lpw;            // problems (active lanes) per wave: 64 for big batches, fewer to spread a small batch over all CUs
};

// Addressing: every access is a raw buffer load/store
//     buffer_load_dwordx2 v[dst], v_lane8, s[rsrc], s_elem offen
// with  rsrc  = descriptor whose base is the wave-uniform address of (step t, element 0, the wave's first
//               problem) - rebuilt per step with two scalar adds,
//       s_elem = e*NP*8, a loop-invariant SGPR shared by all arrays,
//       v_lane8 = lane*8, ONE VGPR for the whole kernel.
// The time loop therefore carries no vector address arithmetic and no 64-bit pointer registers
// (flat global_load made hipcc keep one 64-bit induction pointer per element: +92 VGPRs at dx=4).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* ubase) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ubase), 0, 0xffffffff, 0x00020000);
}
struct Lane {
    unsigned v8;  // lane * 8
    size_t NP;
};
// Cache policy of the streaming loads / stores: 2 = nt (non-temporal).  Every byte of a sweep is touched once, and at the
// sizes this kernel is chosen for nothing of it fits a cache; measured on 65 536 problems, dense inputs (scripts/debug/
// tpp_nt_sweep.sh): cartpole gains-only 4.81 -> 4.99 TB/s, full 4.74 -> 4.78; double cartpole 3.89 -> 4.08 / 4.35 -> 4.58
// (nt on the loads alone or the stores alone gives about half of that each).
#ifndef TPP_PIN
#define TPP_PIN 1
#endif
#ifndef TPP_LD_AUX
#define TPP_LD_AUX 2
#endif
#ifndef TPP_ST_AUX
#define TPP_ST_AUX 2
#endif
template <int R>
__device__ __forceinline__ void load_vec(double (&x)[R], const double* ubase, const Lane& ln) {
    const __amdgpu_buffer_rsrc_t r = mk_rsrc(ubase);
#pragma unroll
    for (int e = 0; e < R; ++e)
        x[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, ln.v8, (unsigned)(e * ln.NP * 8), TPP_LD_AUX));
}
template <int R>
__device__ __forceinline__ void store_vec(double* ubase, const Lane& ln, const double (&x)[R]) {
    const __amdgpu_buffer_rsrc_t r = mk_rsrc(ubase);
#pragma unroll
    for (int e = 0; e < R; ++e)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x[e]), r, ln.v8, (unsigned)(e * ln.NP * 8), TPP_ST_AUX);
}

// ubase(arr, E, t): wave-uniform pointer to element 0 of step t for the wave's first problem.
#define UBASE(arr, E, t) ((arr) + ((size_t)(t) * (E)) * NP + p0)

// SC: structured cost (trajopt_backward_io_t::cost_w): Cxx = w[t] diag(gxx), Cuu = diag(guu), Cxu = 0 are formed in
// registers (the zeros fold away at compile time) instead of being loaded - 21 of a cartpole step's 51 loads.
template <int N, int M, bool SC>
__device__ __forceinline__ void load_cost(CostIn<N, M>& s, const TppArgs& a, size_t p0, const Lane& off, int t) {
    const size_t NP = a.NP;
    (void)NP;
    if constexpr (SC) {
        const double w = a.io.cost_w[t];  // wave-uniform: one scalar load, issued a step ahead like the rest
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) s.Cxx[i + N * j] = (i == j) ? w * a.io.cost_gxx[i] : 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < M; ++i) s.Cuu[i + M * j] = (i == j) ? a.io.cost_guu[i] : 0.0;
#pragma unroll
        for (int k = 0; k < N * M; ++k) s.Cxu[k] = 0.0;
    } else {
        load_vec<N * N>(s.Cxx, UBASE(a.io.Cxx, N * N, t), off);
        load_vec<M * M>(s.Cuu, UBASE(a.io.Cuu, M * M, t), off);
        load_vec<N * M>(s.Cxu, UBASE(a.io.Cxu, N * M, t), off);
    }
}
template <int N, int M, bool GN>
__device__ __forceinline__ void load_step(StepIn<N, M>& s, const TppArgs& a, size_t p0, const Lane& off, int t) {
    const size_t NP = a.NP;
    load_vec<N * N>(s.A, UBASE(a.io.A, N * N, t), off);
    load_vec<N * M>(s.B, UBASE(a.io.B, N * M, t), off);
    load_vec<N>(s.cx, UBASE(a.io.cx, N, t), off);
    load_vec<M>(s.cu, UBASE(a.io.cu, M, t), off);
    if constexpr (GN) load_vec<M>(s.ur, UBASE(a.io.uref, M, t), off);
}

// One step of util.cpp:128-161.  Returns false when Quu_reg fails the PD test (util.cpp:145-148);
// in that case Qxx..qu of this step have been stored (as the reference does) and V, v are untouched.
// inputs_dead() is called once the Q-function blocks stand in registers of their own: nothing of `in` / `cs` is read
// from there on, and a caller that keeps ONE set of input registers (below: dx = 6, dense cost) re-loads it for the
// next step at that point.
struct NoReload { __device__ __forceinline__ void operator()() const {} };
template <int N, int M, bool FULL, bool GN, class Dead = NoReload>
__device__ __forceinline__ bool bwd_step(StepIn<N, M>& in, CostIn<N, M>& cs, Dead&& inputs_dead, double (&V)[N * N], double (&v)[N], double lam,
                                         int reg, double& dV0, double& dV1, double (&gmax)[M], const TppArgs& a,
                                         size_t p0, const Lane& off, const Lane& offK, const Lane& offkff, int kmul,
                                         int t) {
    const size_t NP = a.NP;
    (void)NP;
    // W = V A, Qxx = Cxx + A^T W  (util.cpp:128) - a column of W at a time: all of W at once is N * N more live
    // registers at the point where both input sets, V and the cost blocks are live as well (dx = 6 then spilled)
    double Qxx[N * N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double Wj[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(V[i + N * k], in.A[k + N * j], s);
            Wj[i] = s;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = cs.Cxx[i + N * j];
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(in.A[k + N * i], Wj[k], s);
            Qxx[i + N * j] = s;
        }
    }
    // Z = V B
    double Z[N * M];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(V[i + N * k], in.B[k + N * j], s);
            Z[i + N * j] = s;
        }
    // Qxu = Cxu + A^T Z (N x M) ; Qux = Qxu^T   (util.cpp:130)
    double Qxu[N * M];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = cs.Cxu[i + N * j];
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(in.A[k + N * i], Z[k + N * j], s);
            Qxu[i + N * j] = s;
        }
    // Quu = Cuu + B^T Z            (util.cpp:129)
    double Quu[M * M];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double s = cs.Cuu[i + M * j];
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(in.B[k + N * i], Z[k + N * j], s);
            Quu[i + M * j] = s;
        }
    // qx = cx + A^T v, qu = cu + B^T v   (util.cpp:132-133)
    double qx[N], qu[M];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = in.cx[i];
#pragma unroll
        for (int k = 0; k < N; ++k) s = fma(in.A[k + N * i], v[k], s);
        qx[i] = s;
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double s = in.cu[i];
#pragma unroll
        for (int k = 0; k < N; ++k) s = fma(in.B[k + N * i], v[k], s);
        qu[i] = s;
    }
    if constexpr (FULL) {
        // Q-function blocks are stored before the PD test, like the reference (util.cpp:128-133)
        if (a.io.Qxx) store_vec<N * N>(UBASE(a.io.Qxx, N * N, t), off, Qxx);
        if (a.io.Quu) store_vec<M * M>(UBASE(a.io.Quu, M * M, t), off, Quu);
        if (a.io.qx) store_vec<N>(UBASE(a.io.qx, N, t), off, qx);
        if (a.io.qu) store_vec<M>(UBASE(a.io.qu, M, t), off, qu);
        if (a.io.Qux) {
            double Qux[M * N];  // Qux(j, i) = Qxu(i, j), (M x N) column-major
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) Qux[j + M * i] = Qxu[i + N * j];
            store_vec<M * N>(UBASE(a.io.Qux, N * M, t), off, Qux);
        }
    }
    // regularised blocks (util.cpp:135-143)
    double Qxu_r[N * M], Quu_r[M * M];
#pragma unroll
    for (int k = 0; k < N * M; ++k) Qxu_r[k] = Qxu[k];
#pragma unroll
    for (int k = 0; k < M * M; ++k) Quu_r[k] = Quu[k];
    if (reg == 2) {
        // (V + lam I) B = Z + lam B
#pragma unroll
        for (int j = 0; j < M; ++j) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) s = fma(in.A[k + N * i], in.B[k + N * j], s);
                Qxu_r[i + N * j] = fma(lam, s, Qxu[i + N * j]);
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) s = fma(in.B[k + N * i], in.B[k + N * j], s);
                Quu_r[i + M * j] = fma(lam, s, Quu[i + M * j]);
            }
        }
    } else if (reg == 1) {
#pragma unroll
        for (int i = 0; i < M; ++i) Quu_r[i + M * i] += lam;
    }
    double ur[M];
#pragma unroll
    for (int j = 0; j < M; ++j) ur[j] = GN ? in.ur[j] : 0.0;
    constexpr bool kPin = TPP_PIN && !std::is_same<typename std::decay<Dead>::type, NoReload>::value;
    if constexpr (kPin) __builtin_amdgcn_sched_barrier(0);   // (the re-load neither above the last use of the old values nor below the gains)
    inputs_dead();
    if constexpr (kPin) __builtin_amdgcn_sched_barrier(0);
    // PD test + gains (util.cpp:145-152)
    double K[M * N], kff[M];  // K(j, i): M x N column-major
    if constexpr (M == 1) {
        const double q = Quu_r[0];
        if (!(q - 100.0 * kEps * fabs(q) > 0.0)) return false;
        const double nr = -1.0 / q;
#pragma unroll
        for (int i = 0; i < N; ++i) K[i] = nr * Qxu_r[i];
        kff[0] = nr * qu[0];
    } else {
        double L[M * M];
        if (!sympd_factor<M>(Quu_r, L)) return false;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double col[M];
#pragma unroll
            for (int j = 0; j < M; ++j) col[j] = Qxu_r[i + N * j];
            chol_solve<M>(L, col);
#pragma unroll
            for (int j = 0; j < M; ++j) K[j + M * i] = -col[j];
        }
        double r[M];
#pragma unroll
        for (int j = 0; j < M; ++j) r[j] = qu[j];
        chol_solve<M>(L, r);
#pragma unroll
        for (int j = 0; j < M; ++j) kff[j] = -r[j];
    }
    // Quu kff, Quu K (unregularised Quu: util.cpp:154-160)
    double Qk[M], QK[M * N];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) s = fma(Quu[i + M * k], kff[k], s);
        Qk[i] = s;
    }
#pragma unroll
    for (int c = 0; c < N; ++c)
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) s = fma(Quu[i + M * k], K[k + M * c], s);
            QK[i + M * c] = s;
        }
    // dV += [kff^T qu, 1/2 kff^T Quu kff]   (util.cpp:154)
    {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            s0 = fma(kff[j], qu[j], s0);
            s1 = fma(kff[j], Qk[j], s1);
        }
        dV0 += s0;
        dV1 += 0.5 * s1;
    }
    // v = qx + K^T Quu kff + K^T qu + Qux^T kff   (util.cpp:156-157)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = qx[i];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            s = fma(K[j + M * i], Qk[j] + qu[j], s);
            s = fma(Qxu[i + N * j], kff[j], s);
        }
        v[i] = s;
    }
    // V = Qxx + K^T Quu K + K^T Qux + Qux^T K, then symmetrise   (util.cpp:159-161)
    double W[N * N];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = Qxx[i + N * j];
#pragma unroll
            for (int k = 0; k < M; ++k) {
                s = fma(K[k + M * i], QK[k + M * j] + Qxu[j + N * k], s);
                s = fma(Qxu[i + N * k], K[k + M * j], s);
            }
            W[i + N * j] = s;  // the unsymmetrised V
        }
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) V[i + N * j] = 0.5 * (W[i + N * j] + W[j + N * i]);

    // stores
    // (the lane's bank of the double-buffered gains is part of offK / offkff: trajopt_backward_io_t::out_bank)
    store_vec<M * N>(UBASE(a.io.K, kmul * N * M, t), offK, K);
    store_vec<M>(UBASE(a.io.kff, kmul * M, t), offkff, kff);
    if constexpr (FULL) {
        if (a.io.V) store_vec<N * N>(UBASE(a.io.V, N * N, t), off, V);
        if (a.io.v) store_vec<N>(UBASE(a.io.v, N, t), off, v);
    }
    if constexpr (GN) {
#pragma unroll
        for (int j = 0; j < M; ++j) gmax[j] = nanmax(gmax[j], fabs(kff[j]) / (fabs(ur[j]) + 1.0));
    }
    return true;
}

// DEPTH: how many steps ahead the inputs are fetched.  One (two register sets) hides the HBM latency under a step's
// arithmetic while several waves share a SIMD; a batch of 16k - 100k problems is one or two waves per SIMD, each of
// which then waits ~5 us per step for 23 KB (dense cartpole) with 0.7 us of arithmetic to do - two steps ahead (three
// register sets, the loop unrolled by three) doubles the bytes every wave keeps in flight.
template <int N, int M, bool FULL, bool GN, bool SC, int DEPTH>
__global__ __launch_bounds__(64) void bwd_tpp_kernel(TppArgs a) {
    const size_t p0 = (size_t)blockIdx.x * a.lpw;  // wave-uniform: first problem of this wave
    const unsigned lane = threadIdx.x;
    const size_t p = p0 + lane;
    if (lane >= (unsigned)a.lpw || p >= a.NP) return;
    if (a.io.active && a.io.active[p] == 0) return;
    const size_t NP = a.NP;
    const int T = a.T;
    const Lane off{lane * 8u, NP};  // host guarantees N*N*NP*8 + 512 < 2^32
    const int kmul = a.io.out_bank ? TRAJOPT_GAIN_BANKS : 1;
    const unsigned bank = (unsigned)gains_bank(a.io.out_bank, p);
    const Lane offK{lane * 8u + bank * (unsigned)(N * M * NP * 8), NP}, offkff{lane * 8u + bank * (unsigned)(M * NP * 8), NP};
    double lam = a.io.lmbda[p];
    double dlam = (a.retry && a.io.dlmbda) ? a.io.dlmbda[p] : 1.0;
    int div = -1, retries = 0;
    double dV0 = 0.0, dV1 = 0.0;
    double gmax[M];
    constexpr bool kOneSet = DEPTH == 0;

    for (;;) {
        div = -1;
        dV0 = 0.0;
        dV1 = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) gmax[j] = -INFINITY;
        double V[N * N], v[N];
        {
            // terminal condition (util.cpp:123-124)
            if constexpr (SC) {
                const double w = a.io.cost_w[T];
#pragma unroll
                for (int j = 0; j < N; ++j)
#pragma unroll
                    for (int i = 0; i < N; ++i) V[i + N * j] = (i == j) ? w * a.io.cost_gxx[i] : 0.0;
            } else {
                load_vec<N * N>(V, UBASE(a.io.Cxx, N * N, T), off);
            }
            load_vec<N>(v, UBASE(a.io.cx, N, T), off);
            if constexpr (FULL) {
                if (a.io.V) store_vec<N * N>(UBASE(a.io.V, N * N, T), off, V);
                if (a.io.v) store_vec<N>(UBASE(a.io.v, N, T), off, v);
            }
        }
        // The prefetch is UNCONDITIONAL (index clamped at 0, the last one is a harmless re-load): a load block
        // under `if (t >= 1)` makes hipcc's waitcnt pass merge a no-load path at the join and emit a near-zero
        // vmcnt there, which waits for the prefetch just issued and exposes the whole HBM latency every step.
        StepIn<N, M> s0, s1;
        int t = T - 1;
        auto clamp = [](int x) { return x >= 0 ? x : 0; };
#define TPP_STEP(SET, COST, AFTER) bwd_step<N, M, FULL, GN>(SET, COST, AFTER, V, v, lam, a.reg, dV0, dV1, gmax, a, p0, off, offK, offkff, kmul, t)
        load_step<N, M, GN>(s0, a, p0, off, t);
        if constexpr (kOneSet) {
            // Dense cost at dx = 6: two input sets are 2 x 92 doubles, which with V, Qxx ... does not fit 512 registers -
            // hipcc then parked ~20 doubles of the prefetched set in scratch, each as "load, wait for EVERYTHING in flight,
            // store": the latency of HBM twenty times a step, plus the traffic (+10 % read, 2 x written).  ONE set instead,
            // re-loaded for step t - 1 as soon as step t has no more use for it: the loads fly during the last third of the
            // step's arithmetic (gains, v, V) and are waited for at the top of the next one - the launch is bound by the
            // bytes, not by that wait.
            CostIn<N, M> c;
            load_cost<N, M, SC>(c, a, p0, off, t);
            auto reload = [&]() {
                load_step<N, M, GN>(s0, a, p0, off, clamp(t - 1));
                load_cost<N, M, SC>(c, a, p0, off, clamp(t - 1));
            };
            // V leaves every step exactly symmetric (util.cpp:161) - only the terminal Cxx_T need not be.  The first step
            // runs on the full matrix, the loop carries the 21 distinct entries: 15 loop-carried doubles (and their
            // copies at the back edge) less.
            if (!TPP_STEP(s0, c, reload)) div = t;
            --t;
            if (div < 0) {
                double Vs[N * (N + 1) / 2];
#pragma unroll
                for (int j = 0; j < N; ++j)
#pragma unroll
                    for (int i = j; i < N; ++i) Vs[i * (i + 1) / 2 + j] = V[i + N * j];
                for (; t >= 0; --t) {
                    double Vf[N * N];
#pragma unroll
                    for (int j = 0; j < N; ++j)
#pragma unroll
                        for (int i = 0; i < N; ++i) Vf[i + N * j] = i >= j ? Vs[i * (i + 1) / 2 + j] : Vs[j * (j + 1) / 2 + i];
                    if (!bwd_step<N, M, FULL, GN>(s0, c, reload, Vf, v, lam, a.reg, dV0, dV1, gmax, a, p0, off, offK, offkff, kmul, t)) { div = t; break; }
#pragma unroll
                    for (int j = 0; j < N; ++j)
#pragma unroll
                        for (int i = j; i < N; ++i) Vs[i * (i + 1) / 2 + j] = Vf[i + N * j];
                }
            }
        } else if constexpr (DEPTH == 2) {
            StepIn<N, M> s2;
            CostIn<N, M> c0, c1, c2;
            load_cost<N, M, SC>(c0, a, p0, off, t);
            load_step<N, M, GN>(s1, a, p0, off, clamp(t - 1));
            load_cost<N, M, SC>(c1, a, p0, off, clamp(t - 1));
            while (t >= 0) {
                load_step<N, M, GN>(s2, a, p0, off, clamp(t - 2));
                load_cost<N, M, SC>(c2, a, p0, off, clamp(t - 2));
                if (!TPP_STEP(s0, c0, NoReload{})) { div = t; break; }
                if (--t < 0) break;
                load_step<N, M, GN>(s0, a, p0, off, clamp(t - 2));
                load_cost<N, M, SC>(c0, a, p0, off, clamp(t - 2));
                if (!TPP_STEP(s1, c1, NoReload{})) { div = t; break; }
                if (--t < 0) break;
                load_step<N, M, GN>(s1, a, p0, off, clamp(t - 2));
                load_cost<N, M, SC>(c1, a, p0, off, clamp(t - 2));
                if (!TPP_STEP(s2, c2, NoReload{})) { div = t; break; }
                --t;
            }
        } else {
            CostIn<N, M> c0, c1;
            load_cost<N, M, SC>(c0, a, p0, off, t);
            while (t >= 0) {
                load_step<N, M, GN>(s1, a, p0, off, clamp(t - 1));
                load_cost<N, M, SC>(c1, a, p0, off, clamp(t - 1));
                if (!TPP_STEP(s0, c0, NoReload{})) { div = t; break; }
                if (--t < 0) break;
                load_step<N, M, GN>(s0, a, p0, off, clamp(t - 1));
                load_cost<N, M, SC>(c0, a, p0, off, clamp(t - 1));
                if (!TPP_STEP(s1, c1, NoReload{})) { div = t; break; }
                --t;
            }
        }
#undef TPP_STEP
        if (div < 0 || !a.retry) break;
        // ilqr.py:166-169
        dlam = fmax(dlam * a.sched.mult_lmbda, a.sched.mult_lmbda);
        lam = fmax(lam * dlam, a.sched.min_lmbda);
        // exit every wave reaches: also when lam can no longer grow (max_lmbda = inf, NaN inputs) - the
        // reference's `while not backpass_done` would spin forever there (ilqr.py:162-173)
        if (!(lam <= a.sched.max_lmbda) || !(lam < 1e300) || ++retries >= kMaxLambdaRetries) break;
    }
    a.io.diverge[p] = div;
    a.io.dV[p] = dV0;
    a.io.dV[NP + p] = dV1;
    if (a.retry) {
        a.io.lmbda[p] = lam;
        if (a.io.dlmbda) a.io.dlmbda[p] = dlam;
    }
    if constexpr (GN) {
        if (a.io.g_norm) {
            double g = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) g += gmax[j];
            a.io.g_norm[p] = g / M;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Quadratisation + backward pass in one launch for LARGE batches of dx <= 4, du = 1 models (65 536 cartpoles), lane =
// problem.  As two launches the expansion A, B, cx, cu makes a round trip through HBM - 200 of the 280 bytes the
// quadratisation moves per problem-step and 200 of the 250 the recursion reads (cartpole, structured cost) - and both
// phases are HBM-bound at that size.  Here a lane walks its problem back in time: state and action of step t (fetched a
// step ahead, from the time-blocked candidate plane ref_plane[p] or the dense reference, and written through like
// fused_qb4.hip does) go through the dual-number expansion (objects.py:68-78,114-117) in registers and straight into
// bwd_step().  65 536 problems are 1024 waves: one per SIMD, each a chain of 500 x ~1900 instructions, all of them
// arithmetic the two-launch path issues as well.  Small batches keep fused_qb4.hip (four problems per wave).
struct FusedTppArgs {
    TppArgs t;
    trajopt_env_t env;
    const double *xref, *uref;
    RefView rv;
};

typedef unsigned ft_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ft_st(double x, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ft_v2u, x), r, off, 0, 0);
}
constexpr unsigned kFtNoStore = 0xfffffff8u;

template <class SYS, bool GN>
__global__ __launch_bounds__(64) void fused_tpp_kernel(FusedTppArgs fa) {
    constexpr int N = SYS::N, M = SYS::M;
    static_assert(M == 1 && N >= 2 && N <= 4, "lane-per-problem fused form: dx <= 4, du = 1");
    const TppArgs& a = fa.t;
    const trajopt_env_t& e = fa.env;
    const size_t p0 = (size_t)blockIdx.x * 64;
    const unsigned lane = threadIdx.x;
    const size_t p = p0 + lane;
    if (p >= a.NP) return;
    if (a.io.active && a.io.active[p] == 0) return;
    const size_t NP = a.NP;
    const int T = a.T;
    const Lane off{lane * 8u, NP};
    const int kmul = a.io.out_bank ? TRAJOPT_GAIN_BANKS : 1;
    const unsigned bank = (unsigned)gains_bank(a.io.out_bank, p);
    const Lane offK{lane * 8u + bank * (unsigned)(N * M * NP * 8), NP}, offkff{lane * 8u + bank * (unsigned)(M * NP * 8), NP};
    double lam = a.io.lmbda[p];
    double dlam = (a.retry && a.io.dlmbda) ? a.io.dlmbda[p] : 1.0;
    int div = -1, retries = 0;
    double dV0 = 0.0, dV1 = 0.0;
    double gmax[M];
    // the reference trajectory: time-blocked candidate plane ref_plane[p] (common.h) when that is >= 0, else dense
    int sel = -1;
    if (fa.rv.ref_plane) sel = fa.rv.ref_plane[p];
    const bool refp = sel >= 0;
    const double* xsrc = refp ? fa.rv.cand_state + (size_t)sel * tb_plane(N, T + 1, NP) : fa.xref;
    const double* usrc = refp ? fa.rv.cand_action + (size_t)sel * tb_plane(M, T, NP) : fa.uref;
    auto xi = [&](int t, int i) { return refp ? tb_index(N, t, i, p, NP) : ((size_t)N * t + i) * NP + p; };
    auto ui = [&](int t) { return refp ? tb_index(M, t, 0, p, NP) : (size_t)t * NP + p; };
    // what is read from a plane is written through to the dense arrays (first sweep), with unconditional stores:
    // a lane with nothing to write passes an offset beyond the range (see fused_qb4.hip on the wait-count pass)
    const bool have_out = fa.rv.xref_out != nullptr && fa.rv.uref_out != nullptr;
    const bool through = have_out && refp;
    const __amdgpu_buffer_rsrc_t rxo = __builtin_amdgcn_make_buffer_rsrc(have_out ? fa.rv.xref_out : a.io.kff, 0,
                                                                         have_out ? (unsigned)((size_t)N * (T + 1) * NP * 8) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ruo = __builtin_amdgcn_make_buffer_rsrc(have_out ? fa.rv.uref_out : a.io.kff, 0,
                                                                         have_out ? (unsigned)((size_t)M * T * NP * 8) : 0u, 0x00020000);
    const double sc = e.cost_times_dt ? e.dt : 1.0;
    bool first = true;

    for (;;) {
        div = -1;
        dV0 = 0.0;
        dV1 = 0.0;
        gmax[0] = -INFINITY;
        double V[N * N], v[N];
        {
            // terminal condition (util.cpp:123-124) from the closed-form cost expansion at x_T (objects.py:72-78)
            const double wT = a.io.cost_w[T];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double xT = xsrc[xi(T, i)];
                double g = 0.0;
                if (wT != 0.0) {
                    const double y = ((e.periodic >> i) & 1) ? wrap_angle(xT) : xT;
                    g = (wT * a.io.cost_gxx[i]) * (y - e.goal[i]);
                }
                v[i] = g;
                if (first && through) fa.rv.xref_out[((size_t)N * T + i) * NP + p] = xT;   // x_T is not part of the loop
#pragma unroll
                for (int j = 0; j < N; ++j) V[i + N * j] = (i == j) ? wT * a.io.cost_gxx[i] : 0.0;
            }
        }
        double nx[N], nu, nul, nw;
        auto fetch = [&](int t) {   // unconditional, index clamped
            t = t < 0 ? 0 : t;
            const int tl = t >= 1 ? t - 1 : 0;
#pragma unroll
            for (int i = 0; i < N; ++i) nx[i] = xsrc[xi(t, i)];
            nu = usrc[ui(t)];
            nul = usrc[ui(tl)];
            nw = a.io.cost_w[t];
        };
        fetch(T - 1);
        for (int t = T - 1; t >= 0; --t) {
            double x[N], u[M];
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = nx[i];
            u[0] = nu;
            const double ul = (e.slew_rate && t >= 1) ? nul : 0.0, w = nw;
            fetch(t - 1);
            const bool wt = through && first;
            ft_st(u[0], ruo, wt ? (unsigned)(((size_t)t * NP + p) * 8) : kFtNoStore);
#pragma unroll
            for (int i = 0; i < N; ++i) ft_st(x[i], rxo, wt ? (unsigned)((((size_t)N * t + i) * NP + p) * 8) : kFtNoStore);
            StepIn<N, M> in;
            CostIn<N, M> cin;
            {
                double J[N][N + M];
                SYS::template jacobian_cols<N + M>(e, x, u, 0, J);
#pragma unroll
                for (int k = 0; k < N; ++k)
#pragma unroll
                    for (int i = 0; i < N; ++i) in.A[i + N * k] = J[i][k];
#pragma unroll
                for (int i = 0; i < N; ++i) in.B[i] = J[i][N];
            }
            // cost expansion (objects.py:72-78), closed form for identity features
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double g = 0.0;
                if (w != 0.0) {
                    const double y = ((e.periodic >> i) & 1) ? wrap_angle(x[i]) : x[i];
                    g = (w * (sc * 2.0 * e.gw[i])) * (y - e.goal[i]);
                }
                in.cx[i] = g;
#pragma unroll
                for (int j = 0; j < N; ++j) cin.Cxx[i + N * j] = (i == j) ? w * a.io.cost_gxx[i] : 0.0;
                cin.Cxu[i] = 0.0;
            }
            in.cu[0] = sc * 2.0 * e.uw[0] * (e.slew_rate ? (u[0] - ul) : u[0]);
            cin.Cuu[0] = a.io.cost_guu[0];
            in.ur[0] = u[0];
            if (!bwd_step<N, M, false, GN>(in, cin, NoReload{}, V, v, lam, a.reg, dV0, dV1, gmax, a, p0, off, offK, offkff, kmul, t)) {
                div = t;
                break;
            }
        }
        first = false;
        if (div < 0 || !a.retry) break;
        // ilqr.py:166-169
        dlam = fmax(dlam * a.sched.mult_lmbda, a.sched.mult_lmbda);
        lam = fmax(lam * dlam, a.sched.min_lmbda);
        if (!(lam <= a.sched.max_lmbda) || !(lam < 1e300) || ++retries >= kMaxLambdaRetries) break;
    }
    a.io.diverge[p] = div;
    a.io.dV[p] = dV0;
    a.io.dV[NP + p] = dV1;
    if (a.retry) {
        a.io.lmbda[p] = lam;
        if (a.io.dlmbda) a.io.dlmbda[p] = dlam;
    }
    if constexpr (GN) {
        if (a.io.g_norm) a.io.g_norm[p] = gmax[0];
    }
}

template <class SYS>
static constexpr bool fused_tpp_model() {
    return SYS::kStatic && SYS::M == 1 && SYS::N >= 2 && SYS::N <= 4;
}

struct FusedTppLaunch {
    FusedTppArgs a;
    bool gn, done;
    hipStream_t st;
    template <class SYS>
    void operator()() {
        if constexpr (fused_tpp_model<SYS>()) {
            const dim3 grid((unsigned)((a.t.NP + 63) / 64)), block(64);
            if (gn) hipLaunchKernelGGL((fused_tpp_kernel<SYS, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((fused_tpp_kernel<SYS, false>), grid, block, 0, st, a);
            done = true;
        }
    }
};

// Shapes / sizes the lane-per-problem fused form takes: pendulum, cartpole (identity cost features), batches large
// enough to give most SIMDs a wave (below that the two HBM-bound launches, or fused_qb4.hip, are faster), 32-bit
// offsets inside a slice and inside the write-through arrays.
constexpr int64_t kFusedTppMinProblems = 32768;
bool fused_tpp_supported(const trajopt_env_t* env, int T, int64_t NP) {
    if (env->model != TRAJOPT_ENV_PENDULUM && env->model != TRAJOPT_ENV_CARTPOLE) return false;
    if (env->nb_features != 0 || NP < kFusedTppMinProblems) return false;
    const int n = env->dm_state;
    return (double)TRAJOPT_GAIN_BANKS * n * n * (double)NP * 8.0 + 512.0 < 4294967296.0 &&
           (double)n * (T + 1) * (double)NP * 8.0 < 4294967296.0 - 1024.0;
}

// io: as for trajopt_ilqr_backward_pass_batched with the structured cost set and gains-only outputs; A, B, cx, cu are
// not used (nor written).
int launch_fused_tpp(const trajopt_env_t* env, const trajopt_backward_io_t& io, int reg, int T, int64_t NP, int retry,
                     const trajopt_lambda_schedule_t& sched, const double* xref, const double* uref, const RefView* rv,
                     hipStream_t st) {
    if (!fused_tpp_supported(env, T, NP) || !io.cost_w) return TRAJOPT_ERR_UNSUPPORTED;
    FusedTppLaunch L;
    L.a.t.io = io; L.a.t.sched = sched; L.a.t.NP = (size_t)NP; L.a.t.T = T; L.a.t.reg = reg; L.a.t.retry = retry; L.a.t.lpw = 64;
    L.a.env = *env; L.a.xref = xref; L.a.uref = uref;
    L.a.rv = rv ? *rv : RefView{nullptr, nullptr, nullptr, nullptr, nullptr};
    L.gn = io.uref != nullptr && io.g_norm != nullptr; L.done = false; L.st = st;
    if (!dispatch_model(*env, L) || !L.done) return TRAJOPT_ERR_UNSUPPORTED;
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}


// Problems per wave.  Measured on MI355X (C2, 4096 cartpoles, gains-only): 64 lanes 0.56 ms, 32 / 16 lanes
// 0.67 ms, 8 lanes 0.9 ms - a wave's step costs the same ~350 fp64 issue slots whatever its active lanes and
// more, emptier waves only add contention, so full waves win even when they cover a quarter of the CUs.
static int pick_lpw(size_t) {
    return 64;
}

// Measured (scripts/debug/tpp_depth_sweep.sh, dense cartpole shape, gains-only, TB/s one step / two steps ahead): 8192
// problems 2.73 / 2.86, 16 384 4.42 / 4.36, 65 536 4.42 / 4.38, 131 072 4.27 / 4.25, 262 144 4.64 / 4.74 - the launch is not
// short of bytes in flight (1024 waves x 46 rows of 512 B), so the deeper prefetch is only taken where it measured ahead.
constexpr size_t kDepth2MinProblems = 196608, kDepth2MaxProblems = (size_t)1 << 40;

template <int N, int M, bool SC>
static int launch_nms(const TppArgs& a, hipStream_t st) {
    const bool full = a.io.Qxx || a.io.Qux || a.io.Quu || a.io.qx || a.io.qu || a.io.V || a.io.v;
    const bool gn = a.io.uref != nullptr && a.io.g_norm != nullptr;
    const dim3 block(64);
    const dim3 grid((unsigned)((a.NP + a.lpw - 1) / a.lpw));
    // two steps ahead where three register sets fit (dx <= 4) and a SIMD holds few waves of this launch; none ahead (one
    // set, re-loaded inside the step) where not even two sets fit (dx = 6 with a dense cost)
    constexpr bool kTwoFit = N * N + N * M <= 24, kOneOnly = !SC && 2 * (N * N + N * M) + N * N + M * M + N * M > 100;
    int depth = kOneOnly ? 0 : (kTwoFit && a.NP >= kDepth2MinProblems && a.NP <= kDepth2MaxProblems) ? 2 : 1;
    if (const char* e = getenv("TRAJOPT_TPP_DEPTH")) {   // experiments
        const int want = atoi(e);
        if ((want == 2 && kTwoFit) || want == 1 || (want == 0 && kOneOnly)) depth = want;
    }
#define TPP_GO(D)                                                                                             \
    {                                                                                                         \
        if (full) {                                                                                           \
            if (gn) hipLaunchKernelGGL((bwd_tpp_kernel<N, M, true, true, SC, D>), grid, block, 0, st, a);     \
            else hipLaunchKernelGGL((bwd_tpp_kernel<N, M, true, false, SC, D>), grid, block, 0, st, a);       \
        } else {                                                                                              \
            if (gn) hipLaunchKernelGGL((bwd_tpp_kernel<N, M, false, true, SC, D>), grid, block, 0, st, a);    \
            else hipLaunchKernelGGL((bwd_tpp_kernel<N, M, false, false, SC, D>), grid, block, 0, st, a);      \
        }                                                                                                     \
        TRAJOPT_HIP_CHECK(hipGetLastError());                                                                 \
        return TRAJOPT_OK;                                                                                    \
    }
    if constexpr (kTwoFit) {
        if (depth == 2) TPP_GO(2)
    }
    if constexpr (kOneOnly) {
        if (depth == 0) TPP_GO(0)
    }
    TPP_GO(1)
#undef TPP_GO
}
template <int N, int M>
static int launch_nm(const TppArgs& a, hipStream_t st) {
    return a.io.cost_w ? launch_nms<N, M, true>(a, st) : launch_nms<N, M, false>(a, st);
}

int launch_backward_tpp(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int retry,
                        const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled) {
    TppArgs a;
    a.io = io;
    a.sched = sched;
    a.NP = (size_t)NP;
    a.T = T;
    a.reg = reg;
    a.retry = retry;
    a.lpw = pick_lpw(a.NP);
    *handled = false;
    if (io.out_bank && (double)TRAJOPT_GAIN_BANKS * n * m * (double)NP * 8.0 + 512.0 >= 4294967296.0) return TRAJOPT_OK;   // 32-bit lane offsets
    *handled = true;
    if (n == 2 && m == 1) return launch_nm<2, 1>(a, st);
    if (n == 3 && m == 1) return launch_nm<3, 1>(a, st);
    if (n == 4 && m == 1) return launch_nm<4, 1>(a, st);
    if (n == 4 && m == 2) return launch_nm<4, 2>(a, st);
    if (n == 6 && m == 1) return launch_nm<6, 1>(a, st);
    *handled = false;
    return TRAJOPT_OK;
}

}  // namespace trajopt
