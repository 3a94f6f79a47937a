// Wave-per-problem fallback kernels with run-time dimensions for the two "next" backward passes:
//   max-entropy (GPS) backward pass     /root/reference/trajopt/gps/src/util.cpp:272-374     any dx <= 32, du <= 16
//   belief-space iLQR backward pass     /root/reference/trajopt/bspilqr/src/util.cpp:83-204  any b <= 8, du <= 8
// The reference's Armadillo code takes any dimensions; the lane-per-problem kernels (gps_tpp.hip, bsp_tpp.hip) are
// instantiated for the shapes the reference's own examples use and these kernels take the rest - the role bwd_wg.hip
// plays for the iLQR pass.  One 64-lane workgroup per problem; every matrix of a step lives in LDS (column-major,
// run-time dimensions), the lanes share the elements of each product, the du x du test / inverse runs on lane 0.
// Built for coverage, not speed: PROBLEM_MINOR accesses of one problem are strided.  The two backward passes fetch the
// inputs of step t - 1 into registers while step t computes (gw::Pre) and park them in the LDS images once the step has
// no more use for those; the wide belief-space operands (T, X, V, Z, U, Y) are streamed from global memory as they are used.
#include "gps_common.h"

namespace trajopt {
namespace gw {

constexpr int kW = 64;

struct Acc {   // PROBLEM_MINOR element access of one problem
    size_t NP, p;
};
__device__ __forceinline__ double gl(const double* a, size_t E, int t, size_t e, const Acc& x) {
    return a[((size_t)t * E + e) * x.NP + x.p];
}
__device__ __forceinline__ void g2l(double* dst, const double* a, int E, int t, const Acc& x, int lane) {
    for (int e = lane; e < E; e += kW) dst[e] = gl(a, (size_t)E, t, (size_t)e, x);
}
__device__ __forceinline__ void l2g(double* a, int E, int t, const Acc& x, const double* src, int lane) {
    if (!a) return;
    for (int e = lane; e < E; e += kW) a[((size_t)t * E + e) * x.NP + x.p] = src[e];
}
// A lane's share of an E-element array (elements lane, lane + 64, ...) of ONE step, held in registers: fetched a step
// ahead, parked in the array's LDS image when the step that is computing has no more use for the image.  SLOTS covers the
// largest array the kernel takes; the slots beyond E cost a scalar compare each.
template <int SLOTS>
struct Pre {
    double r[SLOTS];
    __device__ __forceinline__ void fetch(const double* a, int E, int t, const Acc& x, int lane) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            r[s] = 0.0;
            if (kW * s < E) {   // wave-uniform
                const int e = lane + kW * s;
                r[s] = gl(a, (size_t)E, t, (size_t)(e < E ? e : 0), x);
            }
        }
    }
    __device__ __forceinline__ void park(double* dst, int E, int lane) const {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (kW * s < E) {
                const int e = lane + kW * s;
                if (e < E) dst[e] = r[s];
            }
    }
};
// C (r x c) = beta C0 + alpha op(A) op(B), inner dimension k; column-major with leading dimensions lda / ldb of the
// STORED matrices; C must not alias an operand.
__device__ __forceinline__ void mm(double* C, int r, int c, int k, const double* A, int lda, bool tA, const double* B,
                                   int ldb, bool tB, double alpha, const double* C0, double beta, int lane) {
    for (int e = lane; e < r * c; e += kW) {
        const int i = e % r, j = e / r;
        double s = 0.0;
        for (int kk = 0; kk < k; ++kk) {
            const double x = tA ? A[kk + lda * i] : A[i + lda * kk];
            const double y = tB ? B[j + ldb * kk] : B[kk + ldb * j];
            s = fma(x, y, s);
        }
        C[e] = C0 ? fma(beta, C0[e], alpha * s) : alpha * s;
    }
}
__device__ __forceinline__ double wsum(double x) {
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// One lane: Armadillo's is_sympd() rule for the m x m matrix X (as smallmat.h's sympd_factor): tol = 100 eps ||X||_F,
// ||X - X'||_inf / ||X||_inf <= tol, chol(X - tol I) exists.  Y: m x m scratch.
__device__ bool sympd_serial(const double* X, int m, double* Y) {
    double fro = 0.0;
    for (int k = 0; k < m * m; ++k) fro = fma(X[k], X[k], fro);
    const double tol = 100.0 * kEps * sqrt(fro);
    double na = 0.0, nd = 0.0;
    for (int i = 0; i < m; ++i) {
        double ra = 0.0, rd = 0.0;
        for (int j = 0; j < m; ++j) {
            ra += fabs(X[i + m * j]);
            rd += fabs(X[i + m * j] - X[j + m * i]);
        }
        if (ra != ra) return false;
        na = fmax(na, ra);
        nd = fmax(nd, rd);
    }
    if (na != 0.0 && !((nd / na) <= tol)) return false;
    for (int k = 0; k < m * m; ++k) Y[k] = X[k];
    for (int i = 0; i < m; ++i) Y[i + m * i] -= tol;
    for (int j = 0; j < m; ++j) {
        double d = Y[j + m * j];
        for (int k = 0; k < j; ++k) d = fma(-Y[j + m * k], Y[j + m * k], d);
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        Y[j + m * j] = d;
        for (int i = j + 1; i < m; ++i) {
            double s = Y[i + m * j];
            for (int k = 0; k < j; ++k) s = fma(-Y[i + m * k], Y[j + m * k], s);
            Y[i + m * j] = s / d;
        }
    }
    return true;
}

// One lane: Xi = X^-1 and det X by Gauss-Jordan elimination with partial pivoting (the general inv() of the reference);
// false when a pivot is exactly zero or not a number.  A: m x m scratch.
__device__ bool inv_serial(const double* X, int m, double* A, double* Xi, double* det) {
    for (int k = 0; k < m * m; ++k) {
        A[k] = X[k];
        Xi[k] = (k % m == k / m) ? 1.0 : 0.0;
    }
    double d = 1.0;
    for (int k = 0; k < m; ++k) {
        int piv = k;
        for (int i = k + 1; i < m; ++i)
            if (fabs(A[i + m * k]) > fabs(A[piv + m * k])) piv = i;
        const double pv = A[piv + m * k];
        if (pv == 0.0 || pv != pv) { *det = 0.0; return false; }
        if (piv != k) {
            for (int j = 0; j < m; ++j) {
                const double x = A[k + m * j]; A[k + m * j] = A[piv + m * j]; A[piv + m * j] = x;
                const double y = Xi[k + m * j]; Xi[k + m * j] = Xi[piv + m * j]; Xi[piv + m * j] = y;
            }
            d = -d;
        }
        d *= pv;
        const double ip = 1.0 / pv;
        for (int j = 0; j < m; ++j) { A[k + m * j] *= ip; Xi[k + m * j] *= ip; }
        for (int i = 0; i < m; ++i) {
            if (i == k) continue;
            const double f = A[i + m * k];
            for (int j = 0; j < m; ++j) {
                A[i + m * j] = fma(-f, A[k + m * j], A[i + m * j]);
                Xi[i + m * j] = fma(-f, Xi[k + m * j], Xi[i + m * j]);
            }
        }
    }
    *det = d;
    return d == d;
}

}  // namespace gw

// ----------------------------------------------------------------------------------------------------------------
// GPS backward pass, util.cpp:272-374 (algebra as gps_tpp.hip)
// ----------------------------------------------------------------------------------------------------------------
struct GpsGenArgs {
    trajopt_gps_backward_io_t io;
    size_t NP;
    int n, m, T;
};

// SL: register slots per lane of the one-step-ahead input fetch for the dx x dx arrays (64 SL >= dx^2; dx x du and du x du
// arrays likewise) - 0: none, every step's inputs go straight into the LDS images.  Sized for dx = 32 (16 slots an array)
// the fetch registers left one wave per SIMD, and the waves that share a SIMD hide more latency than a lone wave's
// prefetch: dx = 8, du = 4 took 9.1 instead of 3.4 ms.  The launcher picks per shape what measured ahead.
template <int SL>
__global__ __launch_bounds__(64) void gps_generic_kernel(GpsGenArgs a) {
    using namespace gw;
    extern __shared__ __align__(16) double sm[];
    const int n = a.n, m = a.m, T = a.T, nn = n * n, nm = n * m, mm2 = m * m;
    const int lane = threadIdx.x;
    const Acc x{a.NP, (size_t)blockIdx.x};
    const trajopt_gps_backward_io_t& io = a.io;
    double* ptr = sm;
    auto take = [&](int k) { double* r = ptr; ptr += k; return r; };
    double *V = take(nn), *A = take(nn), *Cx = take(nn), *Sg = take(nn), *W = take(nn), *Qxx = take(nn), *Wn = take(nn);
    double *B = take(nm), *Cxu = take(nm), *Z = take(nm), *Qxu = take(nm), *K = take(nm), *Qux = take(nm);
    double *Cuu = take(mm2), *Quu = take(mm2), *Qi = take(mm2), *wk = take(mm2), *sg = take(mm2);
    double *c = take(n), *cx = take(n), *v = take(n), *Vc = take(n), *qx = take(n), *vn = take(n);
    double *cu = take(m), *qu = take(m), *kff = take(m), *flag = take(4);
    g2l(V, io.Cxx, nn, T, x, lane);   // util.cpp:318-320
    g2l(v, io.cx, n, T, x, lane);
    double v0 = gl(io.c0, 1, T, 0, x);
    __syncthreads();
    l2g(io.V, nn, T, x, V, lane);
    l2g(io.v, n, T, x, v, lane);
    if (lane == 0) io.v0[(size_t)T * x.NP + x.p] = v0;
    int status = 0, where = -1;
    // inputs: step T - 1 straight into the LDS images, every later step through registers, one step ahead
    // (dx <= 32, du <= 16: at most 1024 / 512 / 256 elements an array = 16 / 8 / 4 per lane)
    constexpr int kS = SL > 0 ? SL : 1;
    Pre<kS> pA, pCx, pSg, pB, pCxu, pCuu;
    Pre<1> pc, pcx, pcu;
    g2l(A, io.A, nn, T - 1, x, lane); g2l(B, io.B, nm, T - 1, x, lane); g2l(c, io.c, n, T - 1, x, lane);
    g2l(Cx, io.Cxx, nn, T - 1, x, lane); g2l(Cuu, io.Cuu, mm2, T - 1, x, lane); g2l(Cxu, io.Cxu, nm, T - 1, x, lane);
    g2l(cx, io.cx, n, T - 1, x, lane); g2l(cu, io.cu, m, T - 1, x, lane); g2l(Sg, io.sigma_dyn, nn, T - 1, x, lane);
    double c0n = gl(io.c0, 1, T - 1, 0, x), aln = gl(io.alpha, 1, T - 1, 0, x);
    for (int t = T - 1; t >= 0; --t) {
        if constexpr (SL == 0) {
            if (t < T - 1) {
                g2l(A, io.A, nn, t, x, lane); g2l(B, io.B, nm, t, x, lane); g2l(c, io.c, n, t, x, lane);
                g2l(Cx, io.Cxx, nn, t, x, lane); g2l(Cuu, io.Cuu, mm2, t, x, lane); g2l(Cxu, io.Cxu, nm, t, x, lane);
                g2l(cx, io.cx, n, t, x, lane); g2l(cu, io.cu, m, t, x, lane); g2l(Sg, io.sigma_dyn, nn, t, x, lane);
                c0n = gl(io.c0, 1, t, 0, x);
                aln = gl(io.alpha, 1, t, 0, x);
            }
        }
        const double c0t = c0n, al = aln, na = -1.0 / al;
        __syncthreads();
        if constexpr (SL > 0) {   // unconditional (the last one re-loads step 0)
            const int tn = t >= 1 ? t - 1 : 0;
            pA.fetch(io.A, nn, tn, x, lane); pB.fetch(io.B, nm, tn, x, lane); pc.fetch(io.c, n, tn, x, lane);
            pCx.fetch(io.Cxx, nn, tn, x, lane); pCuu.fetch(io.Cuu, mm2, tn, x, lane); pCxu.fetch(io.Cxu, nm, tn, x, lane);
            pcx.fetch(io.cx, n, tn, x, lane); pcu.fetch(io.cu, m, tn, x, lane); pSg.fetch(io.sigma_dyn, nn, tn, x, lane);
            c0n = gl(io.c0, 1, tn, 0, x);
            aln = gl(io.alpha, 1, tn, 0, x);
        }
        mm(W, n, n, n, V, n, false, A, n, false, 1.0, nullptr, 0.0, lane);
        mm(Z, n, m, n, V, n, false, B, n, false, 1.0, nullptr, 0.0, lane);
        mm(Vc, n, 1, n, V, n, false, c, n, false, 1.0, nullptr, 0.0, lane);
        __syncthreads();
        mm(Qxx, n, n, n, A, n, true, W, n, false, na, Cx, na, lane);     // :324
        mm(Qxu, n, m, n, A, n, true, Z, n, false, na, Cxu, na, lane);    // :326 (Qux = Qxu')
        mm(Quu, m, m, n, B, n, true, Z, n, false, na, Cuu, na, lane);    // :325
        for (int i = lane; i < n + m; i += kW) {                          // :328-329
            const bool isx = i < n;
            const double* F = isx ? A + n * i : B + n * (i - n);
            double s = isx ? cx[i] : cu[i - n];
            for (int k = 0; k < n; ++k) s = fma(F[k], fma(2.0, Vc[k], v[k]), s);
            if (isx) qx[i] = na * s; else qu[i - n] = na * s;
        }
        double part = 0.0;                                               // :330-331
        for (int e = lane; e < nn; e += kW) part = fma(V[e % n + n * (e / n)], Sg[e / n + n * (e % n)], part);
        for (int i = lane; i < n; i += kW) part = fma(c[i], Vc[i] + v[i], part);
        const double q0 = na * (c0t + v0 + wsum(part));
        __syncthreads();
        // the input images are dead: step t - 1 moves in (read behind the barrier at the top of the next step)
        if constexpr (SL > 0) {
        pA.park(A, nn, lane); pB.park(B, nm, lane); pc.park(c, n, lane);
        pCx.park(Cx, nn, lane); pCuu.park(Cuu, mm2, lane); pCxu.park(Cxu, nm, lane);
        pcx.park(cx, n, lane); pcu.park(cu, m, lane); pSg.park(Sg, nn, lane);
        }
        for (int e = lane; e < nm; e += kW) Qux[(e / n) + m * (e % n)] = Qxu[e];   // (j, i) <- (i, j)
        l2g(io.Qxx, nn, t, x, Qxx, lane); l2g(io.Quu, mm2, t, x, Quu, lane);
        l2g(io.qx, n, t, x, qx, lane); l2g(io.qu, m, t, x, qu, lane);
        if (io.q0 && lane == 0) io.q0[(size_t)t * x.NP + x.p] = q0;
        if (lane == 0) {   // util.cpp:333-338: a positive definite Quu stops the sweep; inv() of a singular one throws
            const bool pd = sympd_serial(Quu, m, wk);
            double det = 0.0;
            const bool ok = pd ? true : inv_serial(Quu, m, wk, Qi, &det);
            flag[0] = pd ? 1.0 : 0.0;
            flag[1] = ok ? 1.0 : 0.0;
            flag[2] = det;
        }
        __syncthreads();
        l2g(io.Qux, nm, t, x, Qux, lane);
        if (flag[0] != 0.0) { status = 1; where = t; break; }
        if (flag[1] == 0.0) { status = 2; where = t; break; }
        const double det = flag[2];
        mm(K, m, n, m, Qi, m, false, Qxu, n, true, -1.0, nullptr, 0.0, lane);   // :339
        for (int j = lane; j < m; j += kW) {                                      // :340
            double s = 0.0;
            for (int k = 0; k < m; ++k) s = fma(Qi[j + m * k], qu[k], s);
            kff[j] = -0.5 * s;
        }
        for (int e = lane; e < mm2; e += kW) sg[e] = -0.25 * (Qi[e] + Qi[(e / m) + m * (e % m)]);   // :342-343
        __syncthreads();
        mm(Wn, n, n, m, Qxu, n, false, K, m, false, -al, Qxx, -al, lane);          // :348
        for (int i = lane; i < n; i += kW) {                                       // :351
            double s = qx[i];
            for (int k = 0; k < m; ++k) s = fma(2.0 * Qxu[i + n * k], kff[k], s);
            vn[i] = -al * s;
        }
        double qk = 0.0;
        for (int k = 0; k < m; ++k) qk = fma(qu[k], kff[k], qk);
        double det2 = det;                                                         // det(-2 Quu)
        for (int k = 0; k < m; ++k) det2 *= -2.0;
        v0 = -al * (0.5 * qk + q0 + 0.5 * (m * 1.8378770664093453 - log(det2)));  // :352-353, log(2 pi)
        __syncthreads();
        for (int e = lane; e < nn; e += kW) V[e] = 0.5 * (Wn[e] + Wn[(e / n) + n * (e % n)]);   // :349
        for (int i = lane; i < n; i += kW) v[i] = vn[i];
        __syncthreads();
        l2g(io.K, nm, t, x, K, lane); l2g(io.kff, m, t, x, kff, lane); l2g(io.sigma_ctl, mm2, t, x, sg, lane);
        l2g(io.V, nn, t, x, V, lane); l2g(io.v, n, t, x, v, lane);
        if (lane == 0) io.v0[(size_t)t * x.NP + x.p] = v0;
        __syncthreads();
    }
    if (lane == 0) {
        io.diverge[x.p] = where;
        if (io.status) io.status[x.p] = status;
    }
}

bool gps_generic_shape_ok(int n, int m) { return n >= 1 && m >= 1 && n <= 32 && m <= 16; }

int launch_gps_backward_generic(const trajopt_gps_backward_io_t& io, int n, int m, int T, int64_t NP, hipStream_t st) {
    if (!gps_generic_shape_ok(n, m)) return TRAJOPT_ERR_UNSUPPORTED;
    GpsGenArgs a;
    a.io = io; a.NP = (size_t)NP; a.n = n; a.m = m; a.T = T;
    const size_t lds = sizeof(double) * (size_t)(7 * n * n + 6 * n * m + 5 * m * m + 6 * n + 3 * m + 4);
    const int big = n > m ? n : m;
#define GW_GO(SL)                                                                                                          \
    {                                                                                                                      \
        TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)gps_generic_kernel<SL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(gps_generic_kernel<SL>, dim3((unsigned)NP), dim3(64), lds, st, a);                              \
    }
    // Measured (MI355X, T = 100, 4096 problems; no fetch ahead / with): dx = 8, du = 4 (one slot) 3.39 / 4.07 ms - four waves
    // a SIMD without it, three with: the fourth wave hides more than the fetch; dx = 12, du = 4 (four slots) 7.46 / 5.58 ms.
    if (big > 8 && big <= 16) GW_GO(4)      // every array within 256 elements
    else GW_GO(0)
#undef GW_GO
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// The other GPS functions for any dimensions (algebra as gps_ops.hip): forward_pass util.cpp:195-269, augment_cost
// :136-193, kl_divergence :83-121 (one wave per problem / per (problem, step)), quad_expectation :123-134 (one thread
// per problem, straight from global memory).
// ----------------------------------------------------------------------------------------------------------------
namespace gw {
// One lane: inv_sympd() as Armadillo's Cholesky-based routine reads its argument - from the lower triangle; Xi = X^-1,
// det = det X, false when a Cholesky pivot is not positive (Armadillo throws).  A, Y: m x m scratch.
__device__ bool inv_sympd_serial(const double* X, int m, double* A, double* Y, double* Xi, double* det) {
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) Y[i + m * j] = i >= j ? X[i + m * j] : X[j + m * i];
    bool pd = true;
    for (int j = 0; j < m; ++j) {   // pivots of the Cholesky factorisation (A as work space)
        double d = Y[j + m * j];
        for (int k = 0; k < j; ++k) d = fma(-A[j + m * k], A[j + m * k], d);
        pd = pd && (d > 0.0);
        const double sd = sqrt(d);
        A[j + m * j] = sd;
        for (int i = j + 1; i < m; ++i) {
            double x = Y[i + m * j];
            for (int k = 0; k < j; ++k) x = fma(-A[i + m * k], A[j + m * k], x);
            A[i + m * j] = x / sd;
        }
    }
    inv_serial(Y, m, A, Xi, det);
    return pd;
}
}  // namespace gw

struct GpsFwdGenArgs {
    const double *mu_x0, *sigma_x0, *A, *B, *c, *sigma_dyn, *K, *kff, *sigma_ctl;
    double *mu_x, *sigma_x, *mu_u, *sigma_u, *mu_xu, *sigma_xu;
    size_t NP;
    int n, m, T;
};

__global__ __launch_bounds__(64) void gps_forward_generic_kernel(GpsFwdGenArgs a) {
    using namespace gw;
    extern __shared__ __align__(16) double sm[];
    const int n = a.n, m = a.m, T = a.T, nn = n * n, nm = n * m, mm2 = m * m, W = n + m;
    const int lane = threadIdx.x;
    const Acc x{a.NP, (size_t)blockIdx.x};
    double* ptr = sm;
    auto take = [&](int k) { double* r = ptr; ptr += k; return r; };
    double *mx = take(n), *sx = take(nn), *A = take(nn), *B = take(nm), *c = take(n), *S = take(nn), *K = take(nm);
    double *k = take(m), *Sc = take(mm2), *mu = take(m), *ksx = take(nm), *sxk = take(nm), *raw = take(mm2), *su = take(mm2);
    double *J = take(W * W), *mxu = take(W), *FJ = take(n * W), *sn = take(nn), *mn = take(n);
    auto F = [&](int i, int kk) -> double { return kk < n ? A[i + n * kk] : B[i + n * (kk - n)]; };
    g2l(mx, a.mu_x0, n, 0, x, lane);
    g2l(sx, a.sigma_x0, nn, 0, x, lane);
    __syncthreads();
    l2g(a.mu_x, n, 0, x, mx, lane);
    l2g(a.sigma_x, nn, 0, x, sx, lane);
    for (int t = 0; t < T; ++t) {
        g2l(A, a.A, nn, t, x, lane); g2l(B, a.B, nm, t, x, lane); g2l(c, a.c, n, t, x, lane);
        g2l(S, a.sigma_dyn, nn, t, x, lane); g2l(K, a.K, nm, t, x, lane); g2l(k, a.kff, m, t, x, lane);
        g2l(Sc, a.sigma_ctl, mm2, t, x, lane);
        __syncthreads();
        for (int j = lane; j < m; j += kW) {   // :229
            double acc = 0.0;
            for (int q = 0; q < n; ++q) acc = fma(K[j + m * q], mx[q], acc);
            mu[j] = acc + k[j];
        }
        mm(ksx, m, n, n, K, m, false, sx, n, false, 1.0, nullptr, 0.0, lane);   // K sx
        mm(sxk, n, m, n, sx, n, false, K, m, true, 1.0, nullptr, 0.0, lane);    // sx K'
        __syncthreads();
        mm(raw, m, m, n, ksx, m, false, K, m, true, 1.0, Sc, 1.0, lane);        // :232
        __syncthreads();
        for (int e = lane; e < mm2; e += kW) su[e] = 0.5 * (raw[e] + raw[(e / m) + m * (e % m)]);   // :233
        __syncthreads();
        // J = sym([[sx, sx K'], [K sx, su]])   (:237-239), column-major W x W
        for (int e = lane; e < W * W; e += kW) {
            const int i = e % W, j = e / W;
            double v;
            if (i < n && j < n) v = 0.5 * (sx[i + n * j] + sx[j + n * i]);
            else if (i < n) v = 0.5 * (sxk[i + n * (j - n)] + ksx[(j - n) + m * i]);
            else if (j < n) v = 0.5 * (sxk[j + n * (i - n)] + ksx[(i - n) + m * j]);
            else v = 0.5 * (su[(i - n) + m * (j - n)] + su[(j - n) + m * (i - n)]);
            J[e] = v;
        }
        for (int i = lane; i < W; i += kW) mxu[i] = i < n ? mx[i] : mu[i - n];
        __syncthreads();
        l2g(a.mu_u, m, t, x, mu, lane); l2g(a.sigma_u, mm2, t, x, su, lane);
        l2g(a.mu_xu, W, t, x, mxu, lane); l2g(a.sigma_xu, W * W, t, x, J, lane);
        for (int e = lane; e < n * W; e += kW) {   // F J, F = [A B]
            const int i = e % n, j = e / n;
            double acc = 0.0;
            for (int q = 0; q < W; ++q) acc = fma(F(i, q), J[q + W * j], acc);
            FJ[e] = acc;
        }
        __syncthreads();
        for (int e = lane; e < nn; e += kW) {      // :246-251
            const int i = e % n, j = e / n;
            double acc = 0.0;
            for (int q = 0; q < W; ++q) acc = fma(FJ[i + n * q], F(j, q), acc);
            sn[e] = S[e] + acc;
        }
        for (int i = lane; i < n; i += kW) {
            double acc = 0.0;
            for (int q = 0; q < W; ++q) acc = fma(F(i, q), mxu[q], acc);
            mn[i] = acc + c[i];
        }
        __syncthreads();
        for (int e = lane; e < nn; e += kW) sx[e] = 0.5 * (sn[e] + sn[(e / n) + n * (e % n)]);
        for (int i = lane; i < n; i += kW) mx[i] = mn[i];
        __syncthreads();
        l2g(a.mu_x, n, t + 1, x, mx, lane);
        l2g(a.sigma_x, nn, t + 1, x, sx, lane);
    }
    // last slices (:253-256): [mu_x; 0] and the state block only
    for (int i = lane; i < W; i += kW) mxu[i] = i < n ? mx[i] : 0.0;
    for (int e = lane; e < W * W; e += kW) J[e] = (e % W < n && e / W < n) ? sx[(e % W) + n * (e / W)] : 0.0;
    __syncthreads();
    l2g(a.mu_xu, W, T, x, mxu, lane);
    l2g(a.sigma_xu, W * W, T, x, J, lane);
}

struct GpsAugGenArgs {
    const double *Cxx, *cx, *Cuu, *cu, *Cxu, *c0, *K, *kff, *sigma_ctl, *alpha;
    double *agCxx, *agcx, *agCuu, *agcu, *agCxu, *agc0;
    int32_t* status;
    size_t NP;
    int n, m, T;
};

__global__ __launch_bounds__(64) void gps_augment_generic_kernel(GpsAugGenArgs a) {
    using namespace gw;
    extern __shared__ __align__(16) double sm[];
    const int n = a.n, m = a.m, T = a.T, nn = n * n, nm = n * m, mm2 = m * m, t = blockIdx.y;
    const int lane = threadIdx.x;
    const Acc x{a.NP, (size_t)blockIdx.x};
    double* ptr = sm;
    auto take = [&](int k) { double* r = ptr; ptr += k; return r; };
    double *Cxx = take(nn), *cx = take(n), *Cuu = take(mm2), *cu = take(m), *Cxu = take(nm), *K = take(nm), *k = take(m);
    double *sg = take(mm2), *lam = take(mm2), *w1 = take(mm2), *w2 = take(mm2), *KtL = take(nm), *Lk = take(m), *flag = take(2);
    double *oCxx = take(nn);
    g2l(Cxx, a.Cxx, nn, t, x, lane); g2l(cx, a.cx, n, t, x, lane); g2l(Cuu, a.Cuu, mm2, t, x, lane);
    g2l(cu, a.cu, m, t, x, lane); g2l(Cxu, a.Cxu, nm, t, x, lane);
    double c0 = gl(a.c0, 1, t, 0, x);
    if (t < T) {   // (block-uniform)
        g2l(K, a.K, nm, t, x, lane); g2l(k, a.kff, m, t, x, lane); g2l(sg, a.sigma_ctl, mm2, t, x, lane);
        const double al = gl(a.alpha, 1, t, 0, x), ha = 0.5 * al;
        __syncthreads();
        if (lane == 0) {
            double det = 0.0;
            const bool pd = inv_sympd_serial(sg, m, w1, w2, lam, &det);   // inv_sympd() throws in the reference (:164)
            if (!pd) a.status[x.p] = 1;
            flag[0] = det;
        }
        __syncthreads();
        mm(KtL, n, m, m, K, m, true, lam, m, false, 1.0, nullptr, 0.0, lane);   // K' lam
        for (int j = lane; j < m; j += kW) {
            double acc = 0.0;
            for (int q = 0; q < m; ++q) acc = fma(lam[j + m * q], k[q], acc);
            Lk[j] = acc;
        }
        __syncthreads();
        mm(oCxx, n, n, m, KtL, n, false, K, m, false, ha, Cxx, 1.0, lane);       // :166
        double kLk = 0.0;
        for (int j = 0; j < m; ++j) kLk = fma(k[j], Lk[j], kLk);
        __syncthreads();
        for (int e = lane; e < nn; e += kW) Cxx[e] = oCxx[e];
        for (int e = lane; e < mm2; e += kW) Cuu[e] = fma(ha, lam[e], Cuu[e]);   // :167
        for (int e = lane; e < nm; e += kW) Cxu[e] = fma(-ha, KtL[e], Cxu[e]);   // :168
        for (int i = lane; i < n; i += kW) {                                     // :169
            double acc = 0.0;
            for (int q = 0; q < m; ++q) acc = fma(KtL[i + n * q], k[q], acc);
            cx[i] = fma(al, acc, cx[i]);
        }
        for (int j = lane; j < m; j += kW) cu[j] = fma(-al, Lk[j], cu[j]);       // :170
        c0 = c0 + ha * (m * 1.8378770664093453 + log(flag[0])) + ha * kLk;       // :171-172
    }
    __syncthreads();
    l2g(a.agCxx, nn, t, x, Cxx, lane); l2g(a.agcx, n, t, x, cx, lane); l2g(a.agCuu, mm2, t, x, Cuu, lane);
    l2g(a.agcu, m, t, x, cu, lane); l2g(a.agCxu, nm, t, x, Cxu, lane);
    if (lane == 0) a.agc0[(size_t)t * x.NP + x.p] = c0;
}

struct GpsKlGenArgs {
    const double *pK, *pk, *pS, *qK, *qk, *qS, *mu_x, *sigma_x;
    double* kl;
    int32_t* status;
    size_t NP;
    int n, m, T;
};

__global__ __launch_bounds__(64) void gps_kl_generic_kernel(GpsKlGenArgs a) {
    using namespace gw;
    extern __shared__ __align__(16) double sm[];
    const int n = a.n, m = a.m, nn = n * n, nm = n * m, mm2 = m * m, t = blockIdx.y;
    const int lane = threadIdx.x;
    const Acc x{a.NP, (size_t)blockIdx.x};
    double* ptr = sm;
    auto take = [&](int k) { double* r = ptr; ptr += k; return r; };
    double *dK = take(nm), *dk = take(m), *pS = take(mm2), *qS = take(mm2), *mu = take(n), *sx = take(nn);
    double *lam = take(mm2), *w1 = take(mm2), *w2 = take(mm2), *w3 = take(mm2), *LdK = take(nm), *Ldk = take(m), *flag = take(2);
    for (int e = lane; e < nm; e += kW) dK[e] = gl(a.qK, nm, t, e, x) - gl(a.pK, nm, t, e, x);
    for (int j = lane; j < m; j += kW) dk[j] = -gl(a.qk, m, t, j, x) + gl(a.pk, m, t, j, x);
    g2l(pS, a.pS, mm2, t, x, lane); g2l(qS, a.qS, mm2, t, x, lane); g2l(mu, a.mu_x, n, t, x, lane); g2l(sx, a.sigma_x, nn, t, x, lane);
    __syncthreads();
    if (lane == 0) {
        double detq = 0.0, detp = 0.0;
        const bool pd = inv_sympd_serial(qS, m, w1, w2, lam, &detq);   // :102
        if (!pd) a.status[x.p] = 1;
        inv_serial(pS, m, w1, w3, &detp);                              // det(p_sigma)
        flag[0] = detq;
        flag[1] = detp;
    }
    __syncthreads();
    mm(LdK, m, n, m, lam, m, false, dK, m, false, 1.0, nullptr, 0.0, lane);
    for (int j = lane; j < m; j += kW) {
        double acc = 0.0;
        for (int q = 0; q < m; ++q) acc = fma(lam[j + m * q], dk[q], acc);
        Ldk[j] = acc;
    }
    __syncthreads();
    // diff_K = dK' lam dK (n x n); trace(diff_K sx), mu' diff_K mu, mu' dK' lam dk, dk' lam dk, trace(lam pS)
    double part = 0.0;
    for (int e = lane; e < nn; e += kW) {
        const int i = e % n, j = e / n;
        double d = 0.0;
        for (int q = 0; q < m; ++q) d = fma(dK[q + m * i], LdK[q + m * j], d);
        part = fma(0.5 * d, sx[j + n * i], part);
        part = fma(0.5 * mu[i] * d, mu[j], part);
    }
    for (int i = lane; i < n; i += kW) {
        double d = 0.0;
        for (int q = 0; q < m; ++q) d = fma(dK[q + m * i], Ldk[q], d);
        part = fma(-mu[i], d, part);
    }
    for (int j = lane; j < m; j += kW) part = fma(0.5 * dk[j], Ldk[j], part);
    for (int e = lane; e < mm2; e += kW) part = fma(0.5 * lam[(e % m) + m * (e / m)], pS[(e / m) + m * (e % m)], part);
    const double sum = wsum(part);
    if (lane == 0) a.kl[(size_t)t * x.NP + x.p] = 0.5 * log(flag[0] / flag[1]) - 0.5 * m + sum;
}

__global__ __launch_bounds__(64) void gps_quadexp_generic_kernel(const double* mu, const double* sg, const double* Q,
                                                                  const double* q, const double* q0, double* out,
                                                                  size_t NP, int n) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= NP) return;
    double quad = 0.0, lin = 0.0, trc = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) {
            s = fma(mu[(size_t)i * NP + p], Q[((size_t)i + (size_t)n * j) * NP + p], s);
            trc = fma(Q[((size_t)j + (size_t)n * i) * NP + p], sg[((size_t)i + (size_t)n * j) * NP + p], trc);
        }
        quad = fma(s, mu[(size_t)j * NP + p], quad);
        lin = fma(mu[(size_t)j * NP + p], q[(size_t)j * NP + p], lin);
    }
    out[p] = quad + lin + q0[p] + trc;   // :132
}

int launch_gps_forward_generic(const double* mu_x0, const double* sigma_x0, const double* A, const double* B,
                               const double* c, const double* sigma_dyn, const double* K, const double* kff,
                               const double* sigma_ctl, int n, int m, int T, int64_t NP, double* mu_x, double* sigma_x,
                               double* mu_u, double* sigma_u, double* mu_xu, double* sigma_xu, hipStream_t st) {
    if (!gps_generic_shape_ok(n, m)) return TRAJOPT_ERR_UNSUPPORTED;
    GpsFwdGenArgs a{mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, mu_x, sigma_x, mu_u, sigma_u, mu_xu, sigma_xu,
                    (size_t)NP, n, m, T};
    const int W = n + m;
    const size_t lds = sizeof(double) * (size_t)(4 * n * n + 4 * n * m + 3 * m * m + 3 * n + 2 * m + W * W + W + n * W);
    TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)gps_forward_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(gps_forward_generic_kernel, dim3((unsigned)NP), dim3(64), lds, st, a);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

int launch_gps_augment_generic(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                               const double* c0, const double* K, const double* kff, const double* sigma_ctl,
                               const double* alpha, int n, int m, int T, int64_t NP, double* agCxx, double* agcx,
                               double* agCuu, double* agcu, double* agCxu, double* agc0, int32_t* status, hipStream_t st) {
    if (!gps_generic_shape_ok(n, m) || T + 1 > 65535) return TRAJOPT_ERR_UNSUPPORTED;
    GpsAugGenArgs a{Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, agCxx, agcx, agCuu, agcu, agCxu, agc0, status,
                    (size_t)NP, n, m, T};
    const size_t lds = sizeof(double) * (size_t)(2 * n * n + 3 * n * m + 5 * m * m + n + 3 * m + 2);
    TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)gps_augment_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(gps_augment_generic_kernel, dim3((unsigned)NP, (unsigned)(T + 1)), dim3(64), lds, st, a);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

int launch_gps_kl_generic(const double* pK, const double* pk, const double* pS, const double* qK, const double* qk,
                          const double* qS, const double* mu_x, const double* sigma_x, int n, int m, int T, int64_t NP,
                          double* kl, int32_t* status, hipStream_t st) {
    if (!gps_generic_shape_ok(n, m) || T > 65535) return TRAJOPT_ERR_UNSUPPORTED;
    GpsKlGenArgs a{pK, pk, pS, qK, qk, qS, mu_x, sigma_x, kl, status, (size_t)NP, n, m, T};
    const size_t lds = sizeof(double) * (size_t)(n * n + 2 * n * m + 6 * m * m + n + 2 * m + 2);
    TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)gps_kl_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(gps_kl_generic_kernel, dim3((unsigned)NP, (unsigned)T), dim3(64), lds, st, a);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

int launch_gps_quadexp_generic(const double* mu, const double* sg, const double* Q, const double* q, const double* q0,
                               double* out, int n, int64_t NP, hipStream_t st) {
    if (n < 1 || n > 64) return TRAJOPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gps_quadexp_generic_kernel, dim3((unsigned)((NP + 63) / 64)), dim3(64), 0, st, mu, sg, Q, q, q0, out,
                       (size_t)NP, n);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Belief-space iLQR backward pass, bspilqr/src/util.cpp:83-204 (algebra and quirks as bsp_tpp.hip / oracle/bsp_oracle.hpp)
// ----------------------------------------------------------------------------------------------------------------
struct BspGenArgs {
    trajopt_bsp_backward_io_t io;
    size_t NP;
    int b, m, T, reg, tau_mode;
};

__global__ __launch_bounds__(64) void bsp_generic_kernel(BspGenArgs a) {
    using namespace gw;
    extern __shared__ __align__(16) double sm[];
    const int b = a.b, m = a.m, T = a.T, bb = b * b, bm = b * m, mm2 = m * m;
    const int lane = threadIdx.x;
    const Acc x{a.NP, (size_t)blockIdx.x};
    const trajopt_bsp_backward_io_t& io = a.io;
    const double lam = io.lmbda[x.p];
    double* ptr = sm;
    auto take = [&](int k) { double* r = ptr; ptr += k; return r; };
    double *S = take(bb), *F = take(bb), *Q = take(bb), *SF = take(bb), *C = take(bb), *Wn = take(bb);
    double *G = take(bm), *P = take(bm), *SG = take(bm), *Ex = take(bm), *Er = take(bm), *K = take(bm), *DK = take(bm);
    double *R = take(mm2), *D = take(mm2), *Dr = take(mm2), *Di = take(mm2), *wk = take(mm2);
    double *s = take(b), *c = take(b), *sn = take(b), *d = take(m), *kf = take(m), *Dk = take(m);
    double *tau = take(bb), *e = take(bb), *flag = take(4);
    g2l(S, io.Q, bb, T, x, lane);   // :142-144
    g2l(s, io.q, b, T, x, lane);
    g2l(tau, io.p, bb, T, x, lane);
    __syncthreads();
    l2g(io.S, bb, T, x, S, lane);
    l2g(io.s, b, T, x, s, lane);
    l2g(io.tau, bb, T, x, tau, lane);
    double dS0 = 0.0, dS1 = 0.0;
    int where = -1;
    // the dense inputs one step ahead through registers (b, du <= 8: at most 64 elements an array, one per lane)
    Pre<1> pF, pG, pR, pP, pQ;
    g2l(F, io.F, bb, T - 1, x, lane); g2l(G, io.G, bm, T - 1, x, lane); g2l(R, io.R, mm2, T - 1, x, lane);
    g2l(P, io.P, bm, T - 1, x, lane); g2l(Q, io.Q, bb, T - 1, x, lane);
    for (int t = T - 1; t >= 0; --t) {
        __syncthreads();
        {
            const int tn = t >= 1 ? t - 1 : 0;
            pF.fetch(io.F, bb, tn, x, lane); pG.fetch(io.G, bm, tn, x, lane); pR.fetch(io.R, mm2, tn, x, lane);
            pP.fetch(io.P, bm, tn, x, lane); pQ.fetch(io.Q, bb, tn, x, lane);
        }
        mm(SF, b, b, b, S, b, false, F, b, false, 1.0, nullptr, 0.0, lane);
        mm(SG, b, m, b, S, b, false, G, b, false, 1.0, nullptr, 0.0, lane);
        // the linear terms (:152-158): one output per lane, the wide operands (b^2 rows) streamed from global memory
        for (int o = lane; o < b + m + bb; o += kW) {
            double acc;
            if (o < b) {                          // c = q + F's + T'tau + 1/2 X'vec(S)
                acc = gl(io.q, b, t, o, x);
                for (int k = 0; k < b; ++k) acc = fma(F[k + b * o], s[k], acc);
                for (int k = 0; k < bb; ++k)
                    acc = fma(gl(io.T, (size_t)bb * b, t, k + (size_t)bb * o, x), tau[k],
                              fma(0.5 * gl(io.X, (size_t)bb * b, t, k + (size_t)bb * o, x), S[k], acc));
                c[o] = acc;
            } else if (o < b + m) {               // d = r + G's + V'tau + 1/2 Z'vec(S)
                const int j = o - b;
                acc = gl(io.r, m, t, j, x);
                for (int k = 0; k < b; ++k) acc = fma(G[k + b * j], s[k], acc);
                for (int k = 0; k < bb; ++k)
                    acc = fma(gl(io.V, (size_t)bb * m, t, k + (size_t)bb * j, x), tau[k],
                              fma(0.5 * gl(io.Z, (size_t)bb * m, t, k + (size_t)bb * j, x), S[k], acc));
                d[j] = acc;
            } else {                              // e = p + U'tau (tau_mode 0) + 1/2 Y'vec(S)
                const int j = o - b - m;
                acc = gl(io.p, bb, t, j, x);
                for (int k = 0; k < bb; ++k) {
                    if (a.tau_mode == 0) acc = fma(gl(io.U, (size_t)bb * bb, t, k + (size_t)bb * j, x), tau[k], acc);
                    acc = fma(0.5 * gl(io.Y, (size_t)bb * bb, t, k + (size_t)bb * j, x), S[k], acc);
                }
                e[j] = acc;
            }
        }
        __syncthreads();
        mm(C, b, b, b, F, b, true, SF, b, false, 1.0, Q, 1.0, lane);     // :148
        mm(D, m, m, b, G, b, true, SG, b, false, 1.0, R, 1.0, lane);     // :149
        mm(Ex, b, m, b, F, b, true, SG, b, false, 1.0, P, 1.0, lane);    // :150, E = Ex'
        __syncthreads();
        // regularised blocks (:160-168): F'(S + lam I)G = F'SG + lam F'G, G'(S + lam I)G = G'SG + lam G'G
        if (a.reg == 2) {
            mm(Er, b, m, b, F, b, true, G, b, false, lam, Ex, 1.0, lane);
            mm(Dr, m, m, b, G, b, true, G, b, false, lam, D, 1.0, lane);
        } else {
            for (int k = lane; k < bm; k += kW) Er[k] = Ex[k];
            for (int k = lane; k < mm2; k += kW) Dr[k] = D[k] + ((k % m == k / m) ? lam : 0.0);
        }
        __syncthreads();
        pF.park(F, bb, lane); pG.park(G, bm, lane); pR.park(R, mm2, lane); pP.park(P, bm, lane); pQ.park(Q, bb, lane);
        if (lane == 0) {   // :170-175
            const bool pd = sympd_serial(Dr, m, wk);
            double det = 0.0;
            const bool ok = pd && inv_serial(Dr, m, wk, Di, &det);
            flag[0] = pd ? 1.0 : 0.0;
            flag[1] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        if (flag[0] == 0.0 || flag[1] == 0.0) { where = t; break; }
        mm(K, m, b, m, Di, m, false, Er, b, true, -1.0, nullptr, 0.0, lane);   // :176, K = -Dinv Ereg, Ereg = Er'
        for (int j = lane; j < m; j += kW) {                                     // :177
            double acc = 0.0;
            for (int k = 0; k < m; ++k) acc = fma(Di[j + m * k], d[k], acc);
            kf[j] = -acc;
        }
        __syncthreads();
        mm(DK, m, b, m, D, m, false, K, m, false, 1.0, nullptr, 0.0, lane);
        for (int j = lane; j < m; j += kW) {
            double acc = 0.0;
            for (int k = 0; k < m; ++k) acc = fma(D[j + m * k], kf[k], acc);
            Dk[j] = acc;
        }
        __syncthreads();
        {   // :179 (every lane the same sums)
            double s0 = 0.0, s1 = 0.0;
            for (int k = 0; k < m; ++k) { s0 = fma(kf[k], d[k], s0); s1 = fma(kf[k], Dk[k], s1); }
            dS0 += s0;
            dS1 += 0.5 * s1;
        }
        for (int i = lane; i < b; i += kW) {                                     // :183-184
            double acc = c[i];
            for (int k = 0; k < m; ++k) acc = fma(K[k + m * i], Dk[k] + d[k], fma(Ex[i + b * k], kf[k], acc));
            sn[i] = acc;
        }
        for (int q = lane; q < bb; q += kW) {                                    // :186-187
            const int i = q % b, j = q / b;
            double acc = C[q];
            for (int k = 0; k < m; ++k)
                acc = fma(K[k + m * i], DK[k + m * j] + Ex[j + b * k], fma(Ex[i + b * k], K[k + m * j], acc));
            Wn[q] = acc;
        }
        __syncthreads();
        for (int q = lane; q < bb; q += kW) {
            S[q] = 0.5 * (Wn[q] + Wn[(q / b) + b * (q % b)]);                    // :188
            tau[q] = e[q];                                                       // :181
        }
        for (int i = lane; i < b; i += kW) s[i] = sn[i];
        __syncthreads();
        l2g(io.K, bm, t, x, K, lane); l2g(io.kff, m, t, x, kf, lane);
        l2g(io.S, bb, t, x, S, lane); l2g(io.s, b, t, x, s, lane); l2g(io.tau, bb, t, x, tau, lane);
        __syncthreads();
    }
    if (lane == 0) {
        io.dS[x.p] = dS0;
        io.dS[x.NP + x.p] = dS1;
        io.diverge[x.p] = where;
    }
}

bool bsp_generic_shape_ok(int b, int m) { return b >= 1 && m >= 1 && b <= 8 && m <= 8; }

int launch_bsp_backward_generic(const trajopt_bsp_backward_io_t& io, int reg, int tau_mode, int b, int m, int T,
                                int64_t NP, hipStream_t st) {
    if (!bsp_generic_shape_ok(b, m)) return TRAJOPT_ERR_UNSUPPORTED;
    BspGenArgs a;
    a.io = io; a.NP = (size_t)NP; a.b = b; a.m = m; a.T = T; a.reg = reg; a.tau_mode = tau_mode;
    const size_t lds = sizeof(double) * (size_t)(8 * b * b + 7 * b * m + 5 * m * m + 3 * b + 3 * m + 4);
    hipLaunchKernelGGL(bsp_generic_kernel, dim3((unsigned)NP), dim3(64), lds, st, a);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

}  // namespace trajopt
