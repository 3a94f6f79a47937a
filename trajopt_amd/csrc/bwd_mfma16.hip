// Wave-per-problem regularised Riccati backward pass on the fp64 matrix cores, for dx + du <= 16 and du <= 4
// (BASELINE config C4: 12 + 4; also the general path for 4 < dx <= 15).
//
// Row A1 of the hot path: /root/reference/trajopt/ilqr/src/util.cpp:123-162 (+ the lambda retry loop of
// /root/reference/trajopt/ilqr/ilqr.py:162-173).
//
// With F = [A B] (n x d, d = n + m) the whole Q-function Hessian of a step is ONE 16 x 16 tile
//     Q = [Cxx Cxu; Cxu' Cuu] + F' V' F = [Qxx Qxu; Qux Quu]                      (util.cpp:128-130)
// and v_mfma_f64_16x16x4_f64 keeps every operand where the previous instruction left it:
//   * its C/D layout is  D[row = g + 4r][col = c]  (c = lane & 15, g = lane >> 4, r = register 0..3), its B
//     operand of k-step s is  B[k = 4s + g][col = c]  - so D register s IS the B operand of k-step s;
//   * its A operand of k-step s is  A[row = c][k = 4s + g]; for a symmetric matrix held in D layout that is
//     again D register s, and for F' it is the B-layout value of F.
// Hence  P = V' F  (S = ceil(n/4) MFMAs),  Q = C + F' P  (S MFMAs)  and the rank-m update
//     V = Qxx + K' (Quu K + Qux) + Qux' K                                           (util.cpp:159-160)
// (2 MFMAs) run without any data movement between them; V' never leaves the registers.  The m x m PD test
// and the solves for K, kff run redundantly in every lane from an LDS copy of Q (m <= 4, square-root-free
// LDL' with one refined reciprocal per pivot: fp64 divisions and square roots cost 80-115 cycles each here);
// the symmetrisation (util.cpp:161) is one transposed LDS round trip.  Inputs of step t-1 are prefetched
// unconditionally through per-lane pointers while step t computes.
#include <cstdlib>
#include "smallmat.h"

namespace trajopt {

typedef double d4 __attribute__((ext_vector_type(4)));

struct Mf16Args {
    trajopt_backward_io_t io;
    trajopt_lambda_schedule_t sched;
    size_t NP;
    int n, m, T, reg, retry;
    // The lambda ladder in three launches (launch_sm below).  mode 0: the whole retry loop of ilqr.py:162-173 inside the
    // wave.  mode 1: the same, but a problem that has failed rung `inkernel` and may still grow lambda is left PENDING
    // (diverge = -2 - t, lmbda / dlmbda = the state of that rung, dV[slot 0] = INT_MAX as an integer).  The PROBE
    // instantiation (grid.y = j) then runs rung inkernel + 1 + j of every pending problem without storing anything and
    // records the lowest rung that succeeds (atomicMin on that integer).  mode 2: a pending problem resumes the loop at
    // that rung - or, if none of the `rungs` probed ones succeeded, at the last of them lambda's bound allows.
    int mode, inkernel, rungs;
};

// index of element e of step t of problem p in an array with `eps` elements per step and `slices` steps
template <bool MINOR>
__device__ __forceinline__ size_t at(size_t eps, size_t slices, int t, size_t e, size_t p, size_t NP) {
    return MINOR ? (eps * (size_t)t + e) * NP + p : p * eps * slices + eps * (size_t)t + e;
}

// the same for the gains K / kff, which may hold TRAJOPT_GAIN_BANKS banks (trajopt_backward_io_t::out_bank; kmul = their number)
template <bool MINOR>
__device__ __forceinline__ size_t at_gain(size_t eps, size_t slices, int t, size_t e, size_t p, size_t NP, int kmul,
                                          int bank) {
    return MINOR ? (eps * (size_t)(kmul * t + bank) + e) * NP + p
                 : ((size_t)bank * NP + p) * eps * slices + eps * (size_t)t + e;
}

__device__ __forceinline__ double xor_sum(double x) {  // sum over the four 16-lane rows: lanes c, c+16, c+32, c+48
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

template <int M>
__device__ __forceinline__ double pick(const double (&x)[M], int g) {  // x[g] for a lane-dependent g, 0 beyond M
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) r = (g == i) ? x[i] : r;
    return r;
}

// A per-lane input slot: where this lane's element lives at step t, as a pointer that walks backwards in time.
struct Slot {
    const double* ptr;  // element of step t (a valid dummy address when the slot is padding)
    ptrdiff_t step;     // elements per time step (0 for padding)
    bool live;
    template <bool MINOR>
    __device__ __forceinline__ void set(const double* base, size_t eps, size_t slices, int t, size_t e, size_t p,
                                        size_t NP, bool valid) {
        live = valid;
        ptr = valid ? base + at<MINOR>(eps, slices, t, e, p, NP) : base + at<MINOR>(eps, slices, 0, 0, p, NP);
        step = valid ? (ptrdiff_t)(MINOR ? eps * NP : eps) : 0;
    }
    __device__ __forceinline__ double get() const { const double x = *ptr; return live ? x : 0.0; }
    __device__ __forceinline__ void back() { ptr -= step; }
};

constexpr int kLdQ = 17;  // leading dimension of the 16 x 16 LDS tiles (odd: conflict-free rows and columns)

// One wave is the whole workgroup: its LDS accesses execute in program order, so a round trip through LDS needs no
// s_barrier and, above all, no wait for the global loads in flight (hipcc's __syncthreads() drains vmcnt as well, i.e.
// the prefetch of the next step) - only the compiler must keep the order.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// R1: reg == 1 is known at compile time (lambda I on Quu, util.cpp:141-143: the driver's and the reference's default) -
// the reg == 2 machinery (F'F, the sR tile, a branch per element read) then does not exist in the kernel; the other
// instantiation takes any reg at run time.
// SC: structured cost (trajopt_backward_io_t::cost_w): Cxx_t = w[t] diag(gxx), Cuu = diag(guu), Cxu = 0 are formed in
// registers instead of being read - 1664 of the 3328 bytes a quadrotor step reads.
template <int S, int M, bool MINOR, bool FULL, bool R1, bool SC, bool PROBE = false>
__global__ __launch_bounds__(64) void bwd_mfma16_kernel(Mf16Args a) {
    static_assert(!PROBE || !FULL, "a probe stores nothing");
    __shared__ double sQ[16 * kLdQ];  // Q (later the unsymmetrised V), element (row, col) at row + kLdQ * col
    __shared__ double sq[16];         // [qx; qu], later v
    __shared__ double sR[4 * 16];     // reg == 2 only: rows n.. of lam * F'F, element (i, col) at i + 4 * col
    const trajopt_backward_io_t& io = a.io;
    const size_t NP = a.NP, p = blockIdx.x;
    if (io.active && io.active[p] == 0) return;
    const int n = a.n, m = M, d = n + M, T = a.T;
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const bool live = c < n;
    const bool want_gn = !PROBE && io.uref != nullptr && io.g_norm != nullptr;
    const int reg = R1 ? 1 : a.reg;
    // structured cost: this lane's entries of the (diagonal) tile C = diag(w gxx, guu) in D layout, rows g + 4 r
    double scx[4], scu[4];
    if constexpr (SC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = g + 4 * r;
            double gx = 0.0, gu = 0.0;
#pragma unroll
            for (int i = 0; i < TRAJOPT_ENV_MAX_STATE; ++i) gx = (row == c && row == i && i < n) ? io.cost_gxx[i] : gx;
#pragma unroll
            for (int j = 0; j < TRAJOPT_ENV_MAX_ACT; ++j) gu = (row == c && row == n + j && j < M) ? io.cost_guu[j] : gu;
            scx[r] = gx;
            scu[r] = gu;
        }
    }
    const int kmul = io.out_bank ? TRAJOPT_GAIN_BANKS : 1, bank = gains_bank(io.out_bank, p);
    // F in B layout: Fb[s] = F[4s + g][c] is one slot walked with a fixed row offset of 4 rows per k-step
    const ptrdiff_t row4 = MINOR ? (ptrdiff_t)(4 * NP) : 4;

    double lam = io.lmbda[p];
    double dlam = (a.retry && io.dlmbda) ? io.dlmbda[p] : 1.0;
    int div = -1, retries = 0;
    double dV0 = 0.0, dV1 = 0.0, gmax = -INFINITY;
    bool pending = false;
    int* const ladder_slot = reinterpret_cast<int*>(&io.dV[MINOR ? p : 2 * p]);   // (little endian: the low word of dV0)
    auto grow = [&](double& l, double& dl) {   // ilqr.py:166-167; false: lambda has left its bounds, the rung is not run
        dl = fmax(dl * a.sched.mult_lmbda, a.sched.mult_lmbda);
        l = fmax(l * dl, a.sched.min_lmbda);
        return (l <= a.sched.max_lmbda) && (l < 1e300);
    };
    if constexpr (PROBE) {
        if (io.diverge[p] > -2) return;   // not pending
        for (int i = 0; i <= (int)blockIdx.y; ++i)
            if (!grow(lam, dlam)) return;
    } else if (a.mode == 2) {
        if (io.diverge[p] > -2) return;
        const int first_ok = *ladder_slot;
        int adv = 0;
        if (first_ok != 0x7fffffff) {
            adv = first_ok + 1;
        } else {
            double l = lam, dl = dlam;
            for (int i = 1; i <= a.rungs; ++i) {
                if (!grow(l, dl)) break;
                adv = i;
            }
        }
        for (int i = 0; i < adv; ++i) grow(lam, dlam);
        retries = a.inkernel + adv;
    }

    for (;;) {
        div = -1;
        dV0 = dV1 = 0.0;
        gmax = -INFINITY;
        // ---- per-lane input slots, positioned at step T - 1 ------------------------------------------------
        Slot sF, sC[4], sc;
        if (c < n) sF.set<MINOR>(io.A, nn, T, T - 1, g + (size_t)n * c, p, NP, true);
        else sF.set<MINOR>(io.B, nm, T, T - 1, g + (size_t)n * (c < d ? c - n : 0), p, NP, c < d);
        if constexpr (!SC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // C in D layout: C[g + 4r][c], C = [Cxx Cxu; Cxu' Cuu]
            const int row = g + 4 * r;
            if (row < n) {
                if (c < n) sC[r].set<MINOR>(io.Cxx, nn, T + 1, T - 1, row + (size_t)n * c, p, NP, true);
                else sC[r].set<MINOR>(io.Cxu, nm, T + 1, T - 1, row + (size_t)n * (c < d ? c - n : 0), p, NP, c < d);
            } else {
                const int i = row < d ? row - n : 0;
                if (c < n) sC[r].set<MINOR>(io.Cxu, nm, T + 1, T - 1, c + (size_t)n * i, p, NP, row < d);
                else sC[r].set<MINOR>(io.Cuu, mm, T + 1, T - 1, i + (size_t)m * (c < d ? c - n : 0), p, NP, row < d && c < d);
            }
        }
        }
        if (c < n) sc.set<MINOR>(io.cx, n, T + 1, T - 1, c, p, NP, true);
        else sc.set<MINOR>(io.cu, m, T + 1, T - 1, c < d ? c - n : 0, p, NP, c < d);
        bool frow[S];
#pragma unroll
        for (int s = 0; s < S; ++s) frow[s] = sF.live && (4 * s + g < n);

        // ---- terminal condition (util.cpp:123-124) ---------------------------------------------------------
        // Vd is only ever the A operand V'[c][4s + g]; below it is kept exactly symmetric, but the terminal Cxx
        // is taken as given, so it is loaded transposed here.
        d4 Vd;
        double vrow[S];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = g + 4 * r;
            double xt = 0.0;
            if constexpr (SC) {
                xt = io.cost_w[T] * scx[r];
                if (FULL && io.V && row < n && live) io.V[at<MINOR>(nn, T + 1, T, row + (size_t)n * c, p, NP)] = xt;
            } else if (row < n && live) {
                xt = io.Cxx[at<MINOR>(nn, T + 1, T, c + (size_t)n * row, p, NP)];
                if (FULL && io.V) io.V[at<MINOR>(nn, T + 1, T, row + (size_t)n * c, p, NP)] =
                    io.Cxx[at<MINOR>(nn, T + 1, T, row + (size_t)n * c, p, NP)];
            }
            Vd[r] = xt;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int row = 4 * s + g;
            vrow[s] = row < n ? io.cx[at<MINOR>(n, T + 1, T, row, p, NP)] : 0.0;
        }
        if (FULL && io.v && g == 0 && live) io.v[at<MINOR>(n, T + 1, T, c, p, NP)] = io.cx[at<MINOR>(n, T + 1, T, c, p, NP)];

        double Fb[S], Fn[S], qc, qn;
        d4 Cd, Cn;
#pragma unroll
        for (int s = 0; s < S; ++s) { const double x = sF.ptr[frow[s] ? s * row4 : 0]; Fb[s] = frow[s] ? x : 0.0; }
        auto cost_tile = [&](int t, d4& C) {
            if constexpr (SC) {
                const double wt = io.cost_w[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) C[r] = fma(wt, scx[r], scu[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) C[r] = sC[r].get();
            }
        };
        cost_tile(T - 1, Cd);
        qc = sc.get();

        for (int t = T - 1; t >= 0; --t) {
            // unconditional prefetch of step t - 1 (the slots stop walking at step 0: a harmless re-load)
            if (t >= 1) {
                sF.back();
                if constexpr (!SC) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sC[r].back();
                }
                sc.back();
            }
#pragma unroll
            for (int s = 0; s < S; ++s) { const double x = sF.ptr[frow[s] ? s * row4 : 0]; Fn[s] = frow[s] ? x : 0.0; }
            cost_tile(t >= 1 ? t - 1 : 0, Cn);
            qn = sc.get();
            double ur = 0.0;
            if (want_gn && g == 0 && c < m) ur = io.uref[at<MINOR>(m, T, t, c, p, NP)];

            // P = V' F ;  Q = C + F' P   (util.cpp:128-130)
            d4 Pd = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < S; ++s) Pd = __builtin_amdgcn_mfma_f64_16x16x4f64(Vd[s], Fb[s], Pd, 0, 0, 0);
            d4 Qd = Cd;
#pragma unroll
            for (int s = 0; s < S; ++s) Qd = __builtin_amdgcn_mfma_f64_16x16x4f64(Fb[s], Pd[s], Qd, 0, 0, 0);
            d4 Rd = {0.0, 0.0, 0.0, 0.0};  // F' F, rows n.. give B'A and B'B for reg == 2 (util.cpp:135-141)
            if (reg == 2) {
#pragma unroll
                for (int s = 0; s < S; ++s) Rd = __builtin_amdgcn_mfma_f64_16x16x4f64(Fb[s], Fb[s], Rd, 0, 0, 0);
            }
            // q = c + F' v'  (util.cpp:132-133): per-lane partial over this lane's rows, then over the 4 row groups
            double qv = 0.0;
#pragma unroll
            for (int s = 0; s < S; ++s) qv = fma(Fb[s], vrow[s], qv);
            qv = qc + xor_sum(qv);

            wave_lds_fence();  // previous step's readers of sQ / sq are done
#pragma unroll
            for (int r = 0; r < 4; ++r) sQ[(g + 4 * r) + kLdQ * c] = Qd[r];
            if (g == 0) sq[c] = qv;
            if (reg == 2) {  // Q_reg[n + i][:] = Q[n + i][:] + lam (F'F)[n + i][:]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = g + 4 * r;
                    if (row >= n && row < d) sR[(row - n) + 4 * c] = lam * Rd[r];
                }
            }
            if constexpr (FULL) {
                // Q-function blocks are stored before the PD test, like the reference (util.cpp:128-133).
                // Qux = (Cxu + A'V'B)' (util.cpp:130) is the TRANSPOSE of the tile's upper-right block; the
                // lower-left block B'V'A differs from it when the terminal Cxx is not symmetric.
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = g + 4 * r;
                    if (row < n) {
                        if (c < n) {
                            if (io.Qxx) io.Qxx[at<MINOR>(nn, T, t, row + (size_t)n * c, p, NP)] = Qd[r];
                        } else if (c < d) {
                            if (io.Qux) io.Qux[at<MINOR>(nm, T, t, (c - n) + (size_t)m * row, p, NP)] = Qd[r];
                        }
                    } else if (row < d && c >= n && c < d) {
                        if (io.Quu) io.Quu[at<MINOR>(mm, T, t, (row - n) + (size_t)m * (c - n), p, NP)] = Qd[r];
                    }
                }
                if (g == 0) {
                    if (c < n) { if (io.qx) io.qx[at<MINOR>(n, T, t, c, p, NP)] = qv; }
                    else if (c < d) { if (io.qu) io.qu[at<MINOR>(m, T, t, c - n, p, NP)] = qv; }
                }
            }
            wave_lds_fence();

            // ---- gains -------------------------------------------------------------------------------------------
            // Every lane factors the same m x m matrix and solves for kff; lane (c, g) solves column c of K.  What
            // follows is split over the row groups: lane (c, g) forms only row g of Quu K + Qux, Quu kff (that is the
            // MFMA operand it needs) and the sums over i in util.cpp:154-157 become sums over the four groups.
            const int cc = live ? c : 0;
            double qu[M], Kc[M], Quxc[M], L[M * M], id[M];
            {
                double Quu_r[M * M];
#pragma unroll
                for (int j = 0; j < M; ++j)
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        double xr = sQ[(n + i) + kLdQ * (n + j)];
                        if (reg == 2) xr += sR[i + 4 * (n + j)];
                        else if (reg == 1 && i == j) xr += lam;
                        Quu_r[i + M * j] = xr;
                    }
                if (!sympd_ldl<M>(Quu_r, L, id)) {  // wave-uniform
                    div = t;
                    break;
                }
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                qu[i] = sq[n + i];
                Quxc[i] = sQ[cc + kLdQ * (n + i)];  // Qux[i][cc] = Qxu[cc][i]
                Kc[i] = Quxc[i] + (reg == 2 ? sR[i + 4 * cc] : 0.0);
            }
            double kf[M];
#pragma unroll
            for (int i = 0; i < M; ++i) kf[i] = qu[i];
            ldl_solve<M>(L, id, kf);
            ldl_solve<M>(L, id, Kc);
#pragma unroll
            for (int i = 0; i < M; ++i) { kf[i] = -kf[i]; Kc[i] = -Kc[i]; }
            const bool grow = g < M;                    // this lane owns row g of the m-row quantities
            const int gi = grow ? g : 0;
            const double Kg = (live && grow) ? pick<M>(Kc, g) : 0.0;      // K[g][c]:   A operand of K', B operand of K
            const double Qg = (live && grow) ? pick<M>(Quxc, g) : 0.0;    // Qux[g][c]: A operand of Qux'
            const double kfg = grow ? pick<M>(kf, g) : 0.0, qug = grow ? pick<M>(qu, g) : 0.0;
            double M1g = Qg, Qkg = 0.0;                 // (Quu K + Qux)[g][c], (Quu kff)[g]   (unregularised Quu)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const double qgj = sQ[(n + gi) + kLdQ * (n + j)];
                M1g = fma(qgj, Kc[j], M1g);
                Qkg = fma(qgj, kf[j], Qkg);
            }
            M1g = (live && grow) ? M1g : 0.0;
            // dV += [kff'qu, 1/2 kff' Quu kff]  (util.cpp:154);  v = qx + K'(Quu kff + qu) + Qux' kff  (util.cpp:156-157)
            double s0 = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) s0 = fma(kf[i], qu[i], s0);
            dV0 += s0;
            dV1 += 0.5 * xor_sum(grow ? kfg * Qkg : 0.0);
            const double vn = sq[cc] + xor_sum(grow ? fma(Kg, Qkg + qug, Qg * kfg) : 0.0);
            // V = Qxx + K' M1 + Qux' K  (util.cpp:159-160) as two rank-m MFMA updates of the Q tile
            d4 Wd = __builtin_amdgcn_mfma_f64_16x16x4f64(Kg, M1g, Qd, 0, 0, 0);
            Wd = __builtin_amdgcn_mfma_f64_16x16x4f64(Qg, Kg, Wd, 0, 0, 0);

            // gains out
            if constexpr (!PROBE) {
                if (live && g < M) io.K[at_gain<MINOR>(nm, T, t, g + (size_t)m * c, p, NP, kmul, bank)] = Kg;
                if (c == 0 && g < M) io.kff[at_gain<MINOR>(m, T, t, g, p, NP, kmul, bank)] = kfg;
            }
            if (want_gn && g == 0 && c < m) gmax = nanmax(gmax, fabs(pick<M>(kf, c)) / (fabs(ur) + 1.0));

            wave_lds_fence();  // gains have read sQ / sq
#pragma unroll
            for (int r = 0; r < 4; ++r) sQ[(g + 4 * r) + kLdQ * c] = Wd[r];
            if (g == 0) sq[c] = live ? vn : 0.0;
            wave_lds_fence();
            // symmetrise (util.cpp:161) and clear the padding so that it stays finite
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = g + 4 * r;
                const double tr = sQ[c + kLdQ * row];
                const double x = (row < n && live) ? 0.5 * (Wd[r] + tr) : 0.0;
                Vd[r] = x;
                if (FULL && io.V && row < n && live) io.V[at<MINOR>(nn, T + 1, t, row + (size_t)n * c, p, NP)] = x;
            }
#pragma unroll
            for (int s = 0; s < S; ++s) vrow[s] = (4 * s + g < n) ? sq[4 * s + g] : 0.0;
            if (FULL && io.v && g == 0 && live) io.v[at<MINOR>(n, T + 1, t, c, p, NP)] = vn;

#pragma unroll
            for (int s = 0; s < S; ++s) Fb[s] = Fn[s];
            Cd = Cn;
            qc = qn;
        }
        if (div < 0 || !a.retry || PROBE) break;
        if (a.mode == 1 && retries >= a.inkernel && retries + 1 < kMaxLambdaRetries) {
            double l = lam, dl = dlam;
            if (grow(l, dl)) { pending = true; break; }   // the next rung would run: left to the probes (lam, dlam stay this rung's)
        }
        dlam = fmax(dlam * a.sched.mult_lmbda, a.sched.mult_lmbda);  // ilqr.py:166-169
        lam = fmax(lam * dlam, a.sched.min_lmbda);
        // exit every wave reaches: also when lam can no longer grow (max_lmbda = inf, NaN inputs) - the
        // reference's `while not backpass_done` would spin forever there (ilqr.py:162-173)
        if (!(lam <= a.sched.max_lmbda) || !(lam < 1e300) || ++retries >= kMaxLambdaRetries) break;
        wave_lds_fence();
    }
    if constexpr (PROBE) {
        if (div < 0 && lane == 0) atomicMin(ladder_slot, (int)blockIdx.y);
        return;
    }
    if (pending) {
        if (lane == 0) {
            io.diverge[p] = -2 - div;
            io.lmbda[p] = lam;
            if (io.dlmbda) io.dlmbda[p] = dlam;
            *ladder_slot = 0x7fffffff;
        }
        return;
    }

    if (lane == 0) {
        io.diverge[p] = div;
        io.dV[MINOR ? p : 2 * p] = dV0;
        io.dV[MINOR ? NP + p : 2 * p + 1] = dV1;
        if (a.retry) {
            io.lmbda[p] = lam;
            if (io.dlmbda) io.dlmbda[p] = dlam;
        }
    }
    if (want_gn) {
        wave_lds_fence();
        if (g == 0 && c < m) sq[c] = gmax;
        wave_lds_fence();
        if (lane == 0) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += sq[j];
            io.g_norm[p] = s / m;
        }
    }
}

template <int S, int M>
static int launch_sm(const Mf16Args& a, int layout, hipStream_t st) {
    const dim3 grid((unsigned)a.NP), block(64);
    const bool full = a.io.Qxx || a.io.Qux || a.io.Quu || a.io.qx || a.io.qu || a.io.V || a.io.v;
    const bool minor = layout == TRAJOPT_LAYOUT_PROBLEM_MINOR;
    const bool r1 = a.reg == 1, sc = a.io.cost_w != nullptr;
#define TJ_GO(MI, FU) \
    do { \
        if (r1 && sc) hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, MI, FU, true, true>), grid, block, 0, st, a); \
        else if (r1) hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, MI, FU, true, false>), grid, block, 0, st, a); \
        else if (sc) hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, MI, FU, false, true>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, MI, FU, false, false>), grid, block, 0, st, a); \
    } while (0)
    // The ladder in three launches, for the driver's configuration (gains only, reg = 1, structured cost, retries on).
    // Inside one wave the rungs of ilqr.py:162-173 are a chain of sweeps, and the launch lasts as long as its slowest
    // wave: in the early iterations of the C4 workload a handful of problems climb the whole ladder (8 sweeps of ~0.8 ms
    // each at a lone wave's speed) while 4000 waves have long finished - 8.0 ms a launch where one sweep of everybody
    // takes 2.4.  So a wave does at most `in_wave` retries itself (0 by default); what still fails is probed, kRungs rungs at once in
    // as many waves, and resumes at the rung the reference's loop would have stopped at.  Same sweeps on the same
    // numbers: results are those of mode 0 bit for bit (test_lambda_ladder_in_three_launches_...).
    // Measured (one box, scripts/debug/c4_ladder_sweep.sh: backward launch of iterations 1 .. 11, then the bench's C4
    // line): no ladder 8.05 7.96 6.54 6.84 4.41 5.73 3.25 2.63 2.60 2.63 2.55 ms, iteration 8.81 ms; in_wave = 0:
    // 5.75 5.41 4.88 5.35 4.79 5.27 4.27 3.04 2.69 2.90 3.20, iteration 8.30; 1: 8.47; 2: 8.65; 3: 9.04.  The early
    // iterations (what an MPC loop that restarts its solver every control step runs) gain a quarter, the late ones - one
    // retry for a tenth of the problems, each now probed on 8 rungs - pay ~0.3 ms.  TRAJOPT_NO_LADDER=1 keeps mode 0.
    constexpr int kRungs = 8;
    static const bool no_ladder = getenv("TRAJOPT_NO_LADDER") != nullptr;
    static const int in_wave = getenv("TRAJOPT_LADDER_INKERNEL") ? atoi(getenv("TRAJOPT_LADDER_INKERNEL")) : 0;   // (experiments)
    if (a.retry && r1 && sc && !full && a.io.dlmbda && !no_ladder) {
        Mf16Args a1 = a;
        a1.mode = 1; a1.inkernel = in_wave < 0 ? 0 : in_wave; a1.rungs = kRungs;
        Mf16Args a2 = a1;
        a2.mode = 2;
        const dim3 pgrid((unsigned)a.NP, kRungs);
        if (minor) {
            hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, true, false, true, true>), grid, block, 0, st, a1);
            hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, true, false, true, true, true>), pgrid, block, 0, st, a1);
            hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, true, false, true, true>), grid, block, 0, st, a2);
        } else {
            hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, false, false, true, true>), grid, block, 0, st, a1);
            hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, false, false, true, true, true>), pgrid, block, 0, st, a1);
            hipLaunchKernelGGL((bwd_mfma16_kernel<S, M, false, false, true, true>), grid, block, 0, st, a2);
        }
        TRAJOPT_HIP_CHECK(hipGetLastError());
        return TRAJOPT_OK;
    }
    if (minor && full) TJ_GO(true, true);
    else if (minor) TJ_GO(true, false);
    else if (full) TJ_GO(false, true);
    else TJ_GO(false, false);
#undef TJ_GO
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

// handled = false when (n, m) is outside this kernel's range (dx + du <= 16, du <= 4).
int launch_backward_mfma16(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int layout,
                           int retry, const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled) {
    *handled = false;
    if (n + m > 16 || m > 4 || n < 1) return TRAJOPT_OK;
    if (NP > 0x7fffffff) return TRAJOPT_OK;
    Mf16Args a{};
    a.io = io; a.sched = sched; a.NP = (size_t)NP; a.n = n; a.m = m; a.T = T; a.reg = reg; a.retry = retry;
    *handled = true;
    const int S = (n + 3) / 4;
#define TJ_CASE(SS, MM) if (S == SS && m == MM) return launch_sm<SS, MM>(a, layout, st);
    TJ_CASE(1, 1) TJ_CASE(1, 2) TJ_CASE(1, 3) TJ_CASE(1, 4)
    TJ_CASE(2, 1) TJ_CASE(2, 2) TJ_CASE(2, 3) TJ_CASE(2, 4)
    TJ_CASE(3, 1) TJ_CASE(3, 2) TJ_CASE(3, 3) TJ_CASE(3, 4)
    TJ_CASE(4, 1) TJ_CASE(4, 2) TJ_CASE(4, 3)
#undef TJ_CASE
    *handled = false;
    return TRAJOPT_OK;
}

}  // namespace trajopt
