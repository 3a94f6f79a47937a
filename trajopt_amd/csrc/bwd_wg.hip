// Workgroup-per-problem backward passes for any (dx, du) that fits in LDS (dx <= 64, du <= 16).
//
//   VARIANT 0: regularised iLQR/DDP recursion, /root/reference/trajopt/ilqr/src/util.cpp:123-162   (A1)
//   VARIANT 1: "no-1/2, affine-c" TV-LQR sweep, /root/reference/trajopt/riccati/riccati.py:104-124  (A9)
//
// One workgroup owns one problem for the whole horizon.  V (later Qxx, later the new V) stays in LDS
// across steps; A_t and B_t are staged into LDS once per step (they are re-read dx times), every other
// per-step input is read from HBM exactly once where it is consumed, and every output is written once.
// Either batch layout works: with PROBLEM_MAJOR the per-step slices are contiguous (coalesced rows).
// This is the general/correctness path; dx <= 6 uses the thread-per-problem kernels (bwd_tpp.hip).
#include "common.h"

namespace trajopt {

struct WgArgs {
    trajopt_backward_io_t io;       // VARIANT 0
    trajopt_lambda_schedule_t sched;
    const double* c;                // VARIANT 1: affine term (n, T)
    int32_t* status;                // VARIANT 1
    const int32_t* only;            // VARIANT 1, may be NULL: the problems to run (only[p] != 0)
    size_t NP;
    int n, m, T, reg, retry;
};

template <bool MINOR>
struct G {
    size_t NP, p;
    __device__ __forceinline__ size_t operator()(size_t e_total, size_t E_total) const {
        return MINOR ? e_total * NP + p : p * E_total + e_total;
    }
};

// C[r x c] = sum_k op(A)(r,k) * B(k,c) with all operands in LDS, column-major.
// TA: use A^T (A stored k x r).  Each thread handles output elements tid, tid+BS, ...
template <bool TA>
__device__ __forceinline__ double dot_col(const double* A, int lda, const double* B, int ldb, int i, int j, int K) {
    double s = 0.0;
    if (TA) {
        const double* a = A + (size_t)lda * i;
        const double* b = B + (size_t)ldb * j;
        for (int k = 0; k < K; ++k) s = fma(a[k], b[k], s);
    } else {
        const double* b = B + (size_t)ldb * j;
        for (int k = 0; k < K; ++k) s = fma(A[i + (size_t)lda * k], b[k], s);
    }
    return s;
}

template <bool MINOR, int VARIANT>
__global__ __launch_bounds__(256) void bwd_wg_kernel(WgArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n, m = a.m, T = a.T;
    const int tid = threadIdx.x, BS = blockDim.x;
    const size_t p = blockIdx.x;
    if (VARIANT == 0 && a.io.active && a.io.active[p] == 0) return;
    if (VARIANT == 1 && a.only && a.only[p] == 0) return;
    const G<MINOR> g{a.NP, p};
    // gains K / kff: bank io.out_bank[p] when the driver keeps several (trajopt_backward_io_t::out_bank)
    const int kmul = (VARIANT == 0 && a.io.out_bank) ? TRAJOPT_GAIN_BANKS : 1, bank = VARIANT == 0 ? gains_bank(a.io.out_bank, p) : 0;
    auto gK = [&](size_t eps, size_t slices, int t, size_t e) -> size_t {
        return MINOR ? (eps * (size_t)(kmul * t + bank) + e) * a.NP + p
                     : ((size_t)bank * a.NP + p) * eps * slices + eps * (size_t)t + e;
    };
    const int nn = n * n, nm = n * m, mm = m * m;
    const size_t Enn1 = (size_t)nn * (T + 1), Enm1 = (size_t)nm * (T + 1), Emm1 = (size_t)mm * (T + 1);
    const size_t En1 = (size_t)n * (T + 1), Em1 = (size_t)m * (T + 1);
    const size_t Enn = (size_t)nn * T, Enm = (size_t)nm * T, Emm = (size_t)mm * T, En = (size_t)n * T,
                 Em = (size_t)m * T;

    double* sV = sm;            // n*n : V', then Qxx, then V
    double* sA = sV + nn;       // n*n
    double* sW = sA + nn;       // n*n : V'A, then the unsymmetrised V
    double* sB = sW + nn;       // n*m
    double* sZ = sB + nm;       // n*m : V'B
    double* sQxu = sZ + nm;     // n*m
    double* sQxur = sQxu + nm;  // n*m : regularised
    double* sK = sQxur + nm;    // m*n
    double* sQK = sK + nm;      // m*n : Quu K
    double* sQuu = sQK + nm;    // m*m
    double* sX = sQuu + mm;     // m*m : Quu_reg -> factor used for the solves
    double* sY = sX + mm;       // m*m : Quu_reg - tol I -> factor used for the PD test
    double* sv = sY + mm;       // n   : v'
    double* svv = sv + n;       // n   : v' (+ 2 V' c for VARIANT 1)
    double* sqx = svv + n;      // n
    double* squ = sqx + n;      // m
    double* skff = squ + m;     // m
    double* sQk = skff + m;     // m
    double* sc = sQk + m;       // n   : affine c_t
    double* sS = sc + n;        // 8 scalars: [0] tol, [1] ok flag, [2] dV0, [3] dV1
    int* sPiv = reinterpret_cast<int*>(sS + 8);  // m ints (VARIANT 1 pivoting)

    const trajopt_backward_io_t& io = a.io;
    double lam = 0.0, dlam = 1.0;
    if (VARIANT == 0) {
        lam = io.lmbda[p];
        dlam = (a.retry && io.dlmbda) ? io.dlmbda[p] : 1.0;
    }
    int div = -1, retries = 0;
    double gsum = 0.0;  // thread j < m keeps max_t |kff_j| / (|uref_j| + 1)
    double dV0 = 0.0, dV1 = 0.0;

    for (;;) {
        div = -1;
        dV0 = dV1 = 0.0;
        gsum = -INFINITY;
        // terminal condition
        for (int e = tid; e < nn; e += BS) {
            const double x = io.Cxx[g((size_t)nn * T + e, Enn1)];
            sV[e] = x;
            if (io.V) io.V[g((size_t)nn * T + e, Enn1)] = x;
        }
        for (int e = tid; e < n; e += BS) {
            const double x = io.cx[g((size_t)n * T + e, En1)];
            sv[e] = x;
            if (io.v) io.v[g((size_t)n * T + e, En1)] = x;
        }
        __syncthreads();

        for (int t = T - 1; t >= 0; --t) {
            // P0: stage A_t, B_t (and c_t)
            for (int e = tid; e < nn; e += BS) sA[e] = io.A[g((size_t)nn * t + e, Enn)];
            for (int e = tid; e < nm; e += BS) sB[e] = io.B[g((size_t)nm * t + e, Enm)];
            if (VARIANT == 1)
                for (int e = tid; e < n; e += BS) sc[e] = a.c[g((size_t)n * t + e, En)];
            __syncthreads();
            // P1: W = V'A, Z = V'B, vv = v' (+ 2 V' c)
            for (int e = tid; e < nn; e += BS) sW[e] = dot_col<false>(sV, n, sA, n, e % n, e / n, n);
            for (int e = tid; e < nm; e += BS) sZ[e] = dot_col<false>(sV, n, sB, n, e % n, e / n, n);
            for (int e = tid; e < n; e += BS) {
                double s = sv[e];
                if (VARIANT == 1) s = fma(2.0, dot_col<false>(sV, n, sc, n, e, 0, n), s);
                svv[e] = s;
            }
            __syncthreads();
            // P2: Q blocks.  Qxx overwrites V' (dead from here on).
            for (int e = tid; e < nn; e += BS) {
                const double q = io.Cxx[g((size_t)nn * t + e, Enn1)] + dot_col<true>(sA, n, sW, n, e % n, e / n, n);
                sV[e] = q;
                if (VARIANT == 0 && io.Qxx) io.Qxx[g((size_t)nn * t + e, Enn)] = q;
            }
            for (int e = tid; e < nm; e += BS) {
                const int i = e % n, j = e / n;
                // VARIANT 0: (Cxu + A^T V' B)(i,j) (util.cpp:130); VARIANT 1: (Cxu^T + B^T V' A)(j,i) (riccati.py:110)
                const double q = io.Cxu[g((size_t)nm * t + e, Enm1)] +
                                 (VARIANT == 0 ? dot_col<true>(sA, n, sZ, n, i, j, n) : dot_col<true>(sB, n, sW, n, j, i, n));
                sQxu[e] = q;
                double qr = q;
                if (VARIANT == 0 && a.reg == 2) qr = fma(lam, dot_col<true>(sA, n, sB, n, i, j, n), q);
                sQxur[e] = qr;
                if (VARIANT == 0 && io.Qux) io.Qux[g((size_t)nm * t + (j + (size_t)m * i), Enm)] = q;
            }
            for (int e = tid; e < mm; e += BS) {
                const int i = e % m, j = e / m;
                const double q = io.Cuu[g((size_t)mm * t + e, Emm1)] + dot_col<true>(sB, n, sZ, n, i, j, n);
                sQuu[e] = q;
                double qr = q;
                if (VARIANT == 0) {
                    if (a.reg == 2) qr = fma(lam, dot_col<true>(sB, n, sB, n, i, j, n), q);
                    else if (a.reg == 1 && i == j) qr = q + lam;
                    if (io.Quu) io.Quu[g((size_t)mm * t + e, Emm)] = q;
                }
                sX[e] = qr;
            }
            for (int e = tid; e < n; e += BS) {
                const double q = io.cx[g((size_t)n * t + e, En1)] + dot_col<true>(sA, n, svv, n, e, 0, n);
                sqx[e] = q;
                if (VARIANT == 0 && io.qx) io.qx[g((size_t)n * t + e, En)] = q;
            }
            for (int e = tid; e < m; e += BS) {
                const double q = io.cu[g((size_t)m * t + e, Em1)] + dot_col<true>(sB, n, svv, n, e, 0, n);
                squ[e] = q;
                if (VARIANT == 0 && io.qu) io.qu[g((size_t)m * t + e, Em)] = q;
            }
            __syncthreads();

            if (VARIANT == 0) {
                // P3: Armadillo is_sympd (util.cpp:145): tolerance + symmetry check by one thread, then
                // two Cholesky factorisations side by side: X (for the solves) and Y = X - tol I (the test).
                if (tid == 0) {
                    double fro = 0.0;
                    for (int k = 0; k < mm; ++k) fro = fma(sX[k], sX[k], fro);
                    fro = sqrt(fro);
                    const double tol = 100.0 * kEps * fro;
                    double na = 0.0, nd = 0.0;
                    bool bad = false;
                    for (int i = 0; i < m; ++i) {
                        double ra = 0.0, rd = 0.0;
                        for (int j = 0; j < m; ++j) {
                            ra += fabs(sX[i + m * j]);
                            rd += fabs(sX[i + m * j] - sX[j + m * i]);
                        }
                        bad = bad || (ra != ra);
                        na = fmax(na, ra);
                        nd = fmax(nd, rd);
                    }
                    if (!bad && na != 0.0 && !((nd / na) <= tol)) bad = true;
                    sS[0] = tol;
                    sS[1] = bad ? 0.0 : 1.0;
                }
                __syncthreads();
                for (int e = tid; e < mm; e += BS) sY[e] = sX[e] - ((e % m == e / m) ? sS[0] : 0.0);
                __syncthreads();
                for (int j = 0; j < m; ++j) {
                    if (tid == 0) {
                        const double dy = sY[j + m * j];
                        if (!(dy > 0.0)) sS[1] = 0.0;
                        sY[j + m * j] = sqrt(dy);
                        sX[j + m * j] = sqrt(sX[j + m * j]);
                    }
                    __syncthreads();
                    for (int i = j + 1 + tid; i < m; i += BS) {
                        sY[i + m * j] /= sY[j + m * j];
                        sX[i + m * j] /= sX[j + m * j];
                    }
                    __syncthreads();
                    // trailing update of the lower triangle: (i, k), j < k <= i
                    const int r = m - j - 1;
                    for (int e = tid; e < r * r; e += BS) {
                        const int i = j + 1 + e % r, k = j + 1 + e / r;
                        if (k <= i) {
                            sY[i + m * k] = fma(-sY[i + m * j], sY[k + m * j], sY[i + m * k]);
                            sX[i + m * k] = fma(-sX[i + m * j], sX[k + m * j], sX[i + m * k]);
                        }
                    }
                    __syncthreads();
                }
                if (sS[1] == 0.0) {  // uniform across the block
                    div = t;
                    break;
                }
                // P4: K = -X^{-1} Qux_reg (one column per thread), kff = -X^{-1} qu
                for (int c = tid; c <= n; c += BS) {
                    double* out = (c < n) ? (sK + (size_t)m * c) : skff;
                    for (int i = 0; i < m; ++i) {
                        double s = (c < n) ? sQxur[c + n * i] : squ[i];
                        for (int k = 0; k < i; ++k) s = fma(-sX[i + m * k], out[k], s);
                        out[i] = s / sX[i + m * i];
                    }
                    for (int i = m - 1; i >= 0; --i) {
                        double s = out[i];
                        for (int k = i + 1; k < m; ++k) s = fma(-sX[k + m * i], out[k], s);
                        out[i] = s / sX[i + m * i];
                    }
                    for (int i = 0; i < m; ++i) out[i] = -out[i];
                }
                __syncthreads();
                // P5: Quu K, Quu kff (unregularised Quu)
                for (int e = tid; e < nm; e += BS) sQK[e] = dot_col<false>(sQuu, m, sK, m, e % m, e / m, m);
                for (int e = tid; e < m; e += BS) sQk[e] = dot_col<false>(sQuu, m, skff, m, e, 0, m);
                __syncthreads();
                if (tid == 0) {  // util.cpp:154
                    double s0 = 0.0, s1 = 0.0;
                    for (int j = 0; j < m; ++j) {
                        s0 = fma(skff[j], squ[j], s0);
                        s1 = fma(skff[j], sQk[j], s1);
                    }
                    dV0 += s0;
                    dV1 += 0.5 * s1;
                }
                if (io.uref && io.g_norm && tid < m && !(gsum != gsum)) {  // np.max propagates NaN (ilqr.py:176)
                    const double q = fabs(skff[tid]) / (fabs(io.uref[g((size_t)m * t + tid, Em)]) + 1.0);
                    gsum = (q != q) ? q : fmax(gsum, q);
                }
                // P6: v (util.cpp:156-157), unsymmetrised V into sW (util.cpp:159-160)
                for (int i = tid; i < n; i += BS) {
                    double s = sqx[i];
                    for (int j = 0; j < m; ++j) {
                        s = fma(sK[j + m * i], sQk[j] + squ[j], s);
                        s = fma(sQxu[i + n * j], skff[j], s);
                    }
                    sv[i] = s;
                    if (io.v) io.v[g((size_t)n * t + i, En1)] = s;
                }
                for (int e = tid; e < nn; e += BS) {
                    const int i = e % n, j = e / n;
                    double s = sV[e];
                    for (int k = 0; k < m; ++k) {
                        s = fma(sK[k + m * i], sQK[k + m * j] + sQxu[j + n * k], s);
                        s = fma(sQxu[i + n * k], sK[k + m * j], s);
                    }
                    sW[e] = s;
                }
                for (int e = tid; e < nm; e += BS) io.K[gK(nm, T, t, e)] = sK[e];
                for (int e = tid; e < m; e += BS) io.kff[gK(m, T, t, e)] = skff[e];
                __syncthreads();
                // P7: symmetrise (util.cpp:161)
                for (int e = tid; e < nn; e += BS) {
                    const int i = e % n, j = e / n;
                    const double s = 0.5 * (sW[i + n * j] + sW[j + n * i]);
                    sV[e] = s;
                    if (io.V) io.V[g((size_t)nn * t + e, Enn1)] = s;
                }
                __syncthreads();
            } else {
                // VARIANT 1 (riccati.py:118-124): general inverse -> LU with partial pivoting on X = Quu
                for (int k = 0; k < m; ++k) {
                    if (tid == 0) {
                        if (k == 0) sS[1] = 1.0;
                        int pv = k;
                        double best = fabs(sX[k + m * k]);
                        for (int i = k + 1; i < m; ++i) {
                            const double c = fabs(sX[i + m * k]);
                            if (c > best) { best = c; pv = i; }
                        }
                        sPiv[k] = pv;
                        if (best == 0.0 || best != best) sS[1] = -1.0;
                    }
                    __syncthreads();
                    const int pv = sPiv[k];
                    if (pv != k)
                        for (int j = tid; j < m; j += BS) {
                            const double tmp = sX[k + m * j];
                            sX[k + m * j] = sX[pv + m * j];
                            sX[pv + m * j] = tmp;
                        }
                    __syncthreads();
                    for (int i = k + 1 + tid; i < m; i += BS) sX[i + m * k] /= sX[k + m * k];
                    __syncthreads();
                    const int r = m - k - 1;
                    for (int e = tid; e < r * r; e += BS) {
                        const int i = k + 1 + e % r, j = k + 1 + e / r;
                        sX[i + m * j] = fma(-sX[i + m * k], sX[k + m * j], sX[i + m * j]);
                    }
                    __syncthreads();
                }
                if (sS[1] < 0.0) {
                    div = t;
                    break;
                }
                // K = -Quu^{-1} Qux, kff = -1/2 Quu^{-1} qu   (Qux(j,i) = Cxu^T + B^T V' A = Qxu(i,j))
                for (int c = tid; c <= n; c += BS) {
                    double* out = (c < n) ? (sK + (size_t)m * c) : skff;
                    for (int i = 0; i < m; ++i) out[i] = (c < n) ? sQxu[c + n * i] : squ[i];
                    for (int k = 0; k < m; ++k) {
                        const int pv = sPiv[k];
                        if (pv != k) { const double tmp = out[k]; out[k] = out[pv]; out[pv] = tmp; }
                    }
                    for (int i = 0; i < m; ++i) {
                        double s = out[i];
                        for (int k = 0; k < i; ++k) s = fma(-sX[i + m * k], out[k], s);
                        out[i] = s;
                    }
                    for (int i = m - 1; i >= 0; --i) {
                        double s = out[i];
                        for (int k = i + 1; k < m; ++k) s = fma(-sX[i + m * k], out[k], s);
                        out[i] = s / sX[i + m * i];
                    }
                    const double sc_ = (c < n) ? -1.0 : -0.5;
                    for (int i = 0; i < m; ++i) out[i] *= sc_;
                }
                __syncthreads();
                // V = Qxx + Qux^T K ; v = qx + 2 Qux^T kff   (riccati.py:123-124, matrix product form)
                for (int e = tid; e < nn; e += BS) {
                    const int i = e % n, j = e / n;
                    double s = sV[e];
                    for (int k = 0; k < m; ++k) s = fma(sQxu[i + n * k], sK[k + m * j], s);
                    sW[e] = s;
                }
                for (int i = tid; i < n; i += BS) {
                    double s = 0.0;
                    for (int k = 0; k < m; ++k) s = fma(sQxu[i + n * k], skff[k], s);
                    s = fma(2.0, s, sqx[i]);
                    svv[i] = s;  // sv is still read by nobody, but keep the write after the barrier
                }
                for (int e = tid; e < nm; e += BS) io.K[gK(nm, T, t, e)] = sK[e];
                for (int e = tid; e < m; e += BS) io.kff[gK(m, T, t, e)] = skff[e];
                __syncthreads();
                for (int e = tid; e < nn; e += BS) {
                    sV[e] = sW[e];
                    io.V[g((size_t)nn * t + e, Enn1)] = sW[e];
                }
                for (int i = tid; i < n; i += BS) {
                    sv[i] = svv[i];
                    io.v[g((size_t)n * t + i, En1)] = svv[i];
                }
                __syncthreads();
            }
        }
        if (VARIANT == 1) break;
        if (div < 0 || !a.retry) break;
        dlam = fmax(dlam * a.sched.mult_lmbda, a.sched.mult_lmbda);  // ilqr.py:166-169
        lam = fmax(lam * dlam, a.sched.min_lmbda);
        // exit every wave reaches: also when lam can no longer grow (max_lmbda = inf, NaN inputs) - the
        // reference's `while not backpass_done` would spin forever there (ilqr.py:162-173)
        if (!(lam <= a.sched.max_lmbda) || !(lam < 1e300) || ++retries >= kMaxLambdaRetries) break;
        __syncthreads();
    }

    if (VARIANT == 1) {
        if (tid == 0 && a.status) a.status[p] = (div < 0) ? 0 : div + 1;
        return;
    }
    if (tid == 0) {
        io.diverge[p] = div;
        io.dV[g(0, 2)] = dV0;
        io.dV[g(1, 2)] = dV1;
        if (a.retry) {
            io.lmbda[p] = lam;
            if (io.dlmbda) io.dlmbda[p] = dlam;
        }
    }
    if (io.uref && io.g_norm) {
        __syncthreads();
        double* red = sW;  // reuse
        if (tid < m) red[tid] = gsum;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += red[j];
            io.g_norm[p] = s / m;
        }
    }
}

static size_t wg_lds_bytes(int n, int m) {
    const size_t d = (size_t)3 * n * n + (size_t)6 * n * m + (size_t)3 * m * m + (size_t)4 * n + (size_t)3 * m + 8;
    return d * sizeof(double) + (size_t)m * sizeof(int) + 16;
}

static int pick_block(int n) { return n <= 8 ? 64 : (n <= 16 ? 128 : 256); }

int launch_backward_wg(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int layout,
                       int retry, const trajopt_lambda_schedule_t& sched, hipStream_t st) {
    if (n > 64 || m > 16) return TRAJOPT_ERR_UNSUPPORTED;
    WgArgs a{};
    a.io = io; a.sched = sched; a.c = nullptr; a.status = nullptr;
    a.NP = (size_t)NP; a.n = n; a.m = m; a.T = T; a.reg = reg; a.retry = retry;
    const size_t lds = wg_lds_bytes(n, m);
    const dim3 grid((unsigned)NP), block(pick_block(n));
    if (layout == TRAJOPT_LAYOUT_PROBLEM_MINOR) {
        auto k = bwd_wg_kernel<true, 0>;
        if (lds > 48 * 1024) TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
    } else {
        auto k = bwd_wg_kernel<false, 0>;
        if (lds > 48 * 1024) TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
    }
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

int launch_riccati_wg(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                      const double* A, const double* B, const double* c, int n, int m, int T, int64_t NP,
                      int layout, double* V, double* v, double* K, double* kff, int32_t* status, hipStream_t st,
                      const int32_t* only) {
    if (n > 64 || m > 16) return TRAJOPT_ERR_UNSUPPORTED;
    WgArgs a{};
    a.only = only;
    a.io.Cxx = Cxx; a.io.cx = cx; a.io.Cuu = Cuu; a.io.cu = cu; a.io.Cxu = Cxu; a.io.A = A; a.io.B = B;
    a.io.V = V; a.io.v = v; a.io.K = K; a.io.kff = kff;
    a.c = c; a.status = status;
    a.NP = (size_t)NP; a.n = n; a.m = m; a.T = T; a.reg = 0; a.retry = 0;
    const size_t lds = wg_lds_bytes(n, m);
    const dim3 grid((unsigned)NP), block(pick_block(n));
    if (layout == TRAJOPT_LAYOUT_PROBLEM_MINOR) {
        auto k = bwd_wg_kernel<true, 1>;
        if (lds > 48 * 1024) TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
    } else {
        auto k = bwd_wg_kernel<false, 1>;
        if (lds > 48 * 1024) TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
    }
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

}  // namespace trajopt
