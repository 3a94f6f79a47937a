// The other functions of the reference's GPS native module, batched, PROBLEM_MINOR:
//   /root/reference/trajopt/gps/src/util.cpp:83-121   kl_divergence     thread = (problem, step)
//   /root/reference/trajopt/gps/src/util.cpp:123-134  quad_expectation  thread = problem
//   /root/reference/trajopt/gps/src/util.cpp:136-193  augment_cost      thread = (problem, step)
//   /root/reference/trajopt/gps/src/util.cpp:195-269  forward_pass      lane = problem, time-sequential
// One MBGPS dual evaluation (gps/mbgps.py:173-195) is augment_cost -> backward_pass (gps_tpp.hip) -> forward_pass ->
// quad_expectation -> kl_divergence; all five take and leave their arrays in HBM in the same layout.
// Lane = problem everywhere: every access of a wave is one 512-byte row.  Shapes: gps_shape_supported().
#include <vector>

#include "gps_common.h"

namespace trajopt {

struct GpsFwdArgs {
    const double *mu_x0, *sigma_x0, *A, *B, *c, *sigma_dyn, *K, *kff, *sigma_ctl;
    double *mu_x, *sigma_x, *mu_u, *sigma_u, *mu_xu, *sigma_xu;
    size_t NP;
    int T;
};

template <int N, int M>
struct FwdIn { double A[N * N], B[N * M], c[N], S[N * N], K[M * N], k[M], Sc[M * M]; };

template <int N, int M>
__device__ __forceinline__ void fwd_load(FwdIn<N, M>& s, const GpsFwdArgs& a, int t, size_t p) {
    gld<N * N>(s.A, a.A, N * N, t, a.NP, p);
    gld<N * M>(s.B, a.B, N * M, t, a.NP, p);
    gld<N>(s.c, a.c, N, t, a.NP, p);
    gld<N * N>(s.S, a.sigma_dyn, N * N, t, a.NP, p);
    gld<M * N>(s.K, a.K, N * M, t, a.NP, p);
    gld<M>(s.k, a.kff, M, t, a.NP, p);
    gld<M * M>(s.Sc, a.sigma_ctl, M * M, t, a.NP, p);
}

template <int N, int M>
__global__ __launch_bounds__(64) void gps_forward_kernel(GpsFwdArgs a) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x, NP = a.NP;
    if (p >= NP) return;
    constexpr int W = N + M;
    const int T = a.T;
    double mx[N], sx[N * N];
    gld<N>(mx, a.mu_x0, N, 0, NP, p);
    gld<N * N>(sx, a.sigma_x0, N * N, 0, NP, p);
    gst<N>(a.mu_x, N, 0, NP, p, mx);
    gst<N * N>(a.sigma_x, N * N, 0, NP, p, sx);

    auto step = [&](const FwdIn<N, M>& in, int t) {
        double mu[M], ksx[M * N], sxk[N * M], su[M * M];
#pragma unroll
        for (int j = 0; j < M; ++j) {   // :229
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(in.K[j + M * k], mx[k], s);
            mu[j] = s + in.k[j];
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double s = 0.0, r = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    s = fma(in.K[j + M * k], sx[k + N * i], s);   // (K sx)(j, i)
                    r = fma(sx[i + N * k], in.K[j + M * k], r);   // (sx K')(i, j)
                }
                ksx[j + M * i] = s;
                sxk[i + N * j] = r;
            }
        {
            double raw[M * M];
#pragma unroll
            for (int j = 0; j < M; ++j)
#pragma unroll
                for (int i = 0; i < M; ++i) {   // :232
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < N; ++k) s = fma(ksx[i + M * k], in.K[j + M * k], s);
                    raw[i + M * j] = in.Sc[i + M * j] + s;
                }
#pragma unroll
            for (int j = 0; j < M; ++j)
#pragma unroll
                for (int i = 0; i < M; ++i) su[i + M * j] = 0.5 * (raw[i + M * j] + raw[j + M * i]);   // :233
        }
        // J = sym([[sx, sx K'], [K sx, su]])   (:237-239), column-major W x W
        double J[W * W], mxu[W];
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) J[i + W * j] = 0.5 * (sx[i + N * j] + sx[j + N * i]);
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double v = 0.5 * (sxk[i + N * j] + ksx[j + M * i]);
                J[i + W * (N + j)] = v;
                J[(N + j) + W * i] = v;
            }
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < M; ++i) J[(N + i) + W * (N + j)] = 0.5 * (su[i + M * j] + su[j + M * i]);
#pragma unroll
        for (int i = 0; i < N; ++i) mxu[i] = mx[i];
#pragma unroll
        for (int j = 0; j < M; ++j) mxu[N + j] = mu[j];
        gst<M>(a.mu_u, M, t, NP, p, mu);
        gst<M * M>(a.sigma_u, M * M, t, NP, p, su);
        gst<W>(a.mu_xu, W, t, NP, p, mxu);
        gst<W * W>(a.sigma_xu, W * W, t, NP, p, J);
        // sigma_x' = sym(sigma_dyn + (F J) F'),  mu_x' = F mu_xu + c,  F = [A B]   (:246-251)
        auto F = [&](int i, int k) -> double { return k < N ? in.A[i + N * k] : in.B[i + N * (k - N)]; };
        double FJ[N * W], sn[N * N], mn[N];
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < W; ++k) s = fma(F(i, k), J[k + W * j], s);
                FJ[i + N * j] = s;
            }
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < W; ++k) s = fma(FJ[i + N * k], F(j, k), s);
                sn[i + N * j] = in.S[i + N * j] + s;
            }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k) s = fma(F(i, k), mxu[k], s);
            mn[i] = s + in.c[i];
        }
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) sx[i + N * j] = 0.5 * (sn[i + N * j] + sn[j + N * i]);
#pragma unroll
        for (int i = 0; i < N; ++i) mx[i] = mn[i];
        gst<N>(a.mu_x, N, t + 1, NP, p, mx);
        gst<N * N>(a.sigma_x, N * N, t + 1, NP, p, sx);
    };

    FwdIn<N, M> s0, s1;
    fwd_load<N, M>(s0, a, 0, p);
    int t = 0;
    while (t < T) {
        fwd_load<N, M>(s1, a, t + 1 < T ? t + 1 : T - 1, p);
        step(s0, t);
        if (++t >= T) break;
        fwd_load<N, M>(s0, a, t + 1 < T ? t + 1 : T - 1, p);
        step(s1, t);
        ++t;
    }
    // last slices (:253-256): [mu_x; 0] and the state block only
    double mxu[W], J[W * W];
#pragma unroll
    for (int i = 0; i < W; ++i) mxu[i] = i < N ? mx[i < N ? i : 0] : 0.0;
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int i = 0; i < W; ++i) J[i + W * j] = (i < N && j < N) ? sx[(i < N ? i : 0) + N * (j < N ? j : 0)] : 0.0;
    gst<W>(a.mu_xu, W, T, NP, p, mxu);
    gst<W * W>(a.sigma_xu, W * W, T, NP, p, J);
}

struct GpsAugArgs {
    const double *Cxx, *cx, *Cuu, *cu, *Cxu, *c0, *K, *kff, *sigma_ctl, *alpha;
    double *agCxx, *agcx, *agCuu, *agcu, *agCxu, *agc0;
    int32_t* status;
    size_t NP;
    int T;
};

template <int N, int M>
__global__ __launch_bounds__(64) void gps_augment_kernel(GpsAugArgs a) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x, NP = a.NP;
    const int t = blockIdx.y, T = a.T;
    if (p >= NP) return;
    double Cxx[N * N], cx[N], Cuu[M * M], cu[M], Cxu[N * M];
    gld<N * N>(Cxx, a.Cxx, N * N, t, NP, p);
    gld<N>(cx, a.cx, N, t, NP, p);
    gld<M * M>(Cuu, a.Cuu, M * M, t, NP, p);
    gld<M>(cu, a.cu, M, t, NP, p);
    gld<N * M>(Cxu, a.Cxu, N * M, t, NP, p);
    double c0 = a.c0[(size_t)t * NP + p];
    if (t < T) {
        double K[M * N], k[M], sg[M * M], lam[M * M], det;
        gld<M * N>(K, a.K, N * M, t, NP, p);
        gld<M>(k, a.kff, M, t, NP, p);
        gld<M * M>(sg, a.sigma_ctl, M * M, t, NP, p);
        const double al = a.alpha[(size_t)t * NP + p];
        if (!small_inv_sympd<M>(sg, lam, det)) a.status[p] = 1;   // inv_sympd() throws in the reference (:164)
        double KtL[N * M], Lk[M];   // K' lam, lam k
#pragma unroll
        for (int j = 0; j < M; ++j) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < M; ++q) s = fma(K[q + M * i], lam[q + M * j], s);
                KtL[i + N * j] = s;
            }
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < M; ++q) s = fma(lam[j + M * q], k[q], s);
            Lk[j] = s;
        }
        const double ha = 0.5 * al;
        double kLk = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) kLk = fma(k[j], Lk[j], kLk);
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < M; ++q) s = fma(KtL[i + N * q], K[q + M * j], s);
                Cxx[i + N * j] = fma(ha, s, Cxx[i + N * j]);   // :166
            }
#pragma unroll
        for (int i = 0; i < M * M; ++i) Cuu[i] = fma(ha, lam[i], Cuu[i]);   // :167
#pragma unroll
        for (int i = 0; i < N * M; ++i) Cxu[i] = fma(-ha, KtL[i], Cxu[i]);  // :168
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < M; ++q) s = fma(KtL[i + N * q], k[q], s);
            cx[i] = fma(al, s, cx[i]);   // :169
        }
#pragma unroll
        for (int j = 0; j < M; ++j) cu[j] = fma(-al, Lk[j], cu[j]);   // :170
        // log det(2 pi sigma) = M log(2 pi) + log det(sigma)   (:171-172)
        c0 = c0 + ha * (M * 1.8378770664093453 + log(det)) + ha * kLk;
    }
    gst<N * N>(a.agCxx, N * N, t, NP, p, Cxx);
    gst<N>(a.agcx, N, t, NP, p, cx);
    gst<M * M>(a.agCuu, M * M, t, NP, p, Cuu);
    gst<M>(a.agcu, M, t, NP, p, cu);
    gst<N * M>(a.agCxu, N * M, t, NP, p, Cxu);
    a.agc0[(size_t)t * NP + p] = c0;
}

struct GpsKlArgs {
    const double *pK, *pk, *pS, *qK, *qk, *qS, *mu_x, *sigma_x;
    double* kl;
    int32_t* status;
    size_t NP;
    int T;
};

template <int N, int M>
__global__ __launch_bounds__(64) void gps_kl_kernel(GpsKlArgs a) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x, NP = a.NP;
    const int t = blockIdx.y;
    if (p >= NP) return;
    double pK[M * N], qK[M * N], pk[M], qk[M], pS[M * M], qS[M * M], mu[N], sx[N * N];
    gld<M * N>(pK, a.pK, N * M, t, NP, p);
    gld<M * N>(qK, a.qK, N * M, t, NP, p);
    gld<M>(pk, a.pk, M, t, NP, p);
    gld<M>(qk, a.qk, M, t, NP, p);
    gld<M * M>(pS, a.pS, M * M, t, NP, p);
    gld<M * M>(qS, a.qS, M * M, t, NP, p);
    gld<N>(mu, a.mu_x, N, t, NP, p);
    gld<N * N>(sx, a.sigma_x, N * N, t, NP, p);
    double lam[M * M], detq, detp;
    if (!small_inv_sympd<M>(qS, lam, detq)) a.status[p] = 1;   // :102
    if constexpr (M == 1) detp = pS[0];
    else detp = pS[0] * pS[3] - pS[2] * pS[1];
    double dK[M * N], dk[M], LdK[M * N], Ldk[M];
#pragma unroll
    for (int i = 0; i < M * N; ++i) dK[i] = qK[i] - pK[i];
#pragma unroll
    for (int j = 0; j < M; ++j) dk[j] = -qk[j] + pk[j];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < M; ++q) s = fma(lam[j + M * q], dK[q + M * i], s);
            LdK[j + M * i] = s;
        }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < M; ++q) s = fma(lam[j + M * q], dk[q], s);
        Ldk[j] = s;
    }
    // diff_K = dK' lam dK (N x N); trace(diff_K sx), mu' diff_K mu, mu' dK' lam dk, dk' lam dk
    double tr_ks = 0.0, mKm = 0.0, crs = 0.0, dkk = 0.0, tr_lp = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double d = 0.0;
#pragma unroll
            for (int q = 0; q < M; ++q) d = fma(dK[q + M * i], LdK[q + M * j], d);   // diff_K(i, j)
            tr_ks = fma(d, sx[j + N * i], tr_ks);
            mKm = fma(mu[i] * d, mu[j], mKm);
        }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double d = 0.0;
#pragma unroll
        for (int q = 0; q < M; ++q) d = fma(dK[q + M * i], Ldk[q], d);
        crs = fma(mu[i], d, crs);
    }
#pragma unroll
    for (int j = 0; j < M; ++j) dkk = fma(dk[j], Ldk[j], dkk);
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int q = 0; q < M; ++q) tr_lp = fma(lam[i + M * q], pS[q + M * i], tr_lp);
    a.kl[(size_t)t * NP + p] = 0.5 * log(detq / detp) + 0.5 * tr_lp - 0.5 * M + 0.5 * tr_ks + 0.5 * mKm - crs + 0.5 * dkk;
}

template <int N>
__global__ __launch_bounds__(64) void gps_quadexp_kernel(const double* mu_, const double* sg_, const double* Q_,
                                                         const double* q_, const double* q0_, double* out, size_t NP) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= NP) return;
    double mu[N], sg[N * N], Q[N * N], q[N];
    gld<N>(mu, mu_, N, 0, NP, p);
    gld<N * N>(sg, sg_, N * N, 0, NP, p);
    gld<N * N>(Q, Q_, N * N, 0, NP, p);
    gld<N>(q, q_, N, 0, NP, p);
    double quad = 0.0, lin = 0.0, trc = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            s = fma(mu[i], Q[i + N * j], s);
            trc = fma(Q[j + N * i], sg[i + N * j], trc);
        }
        quad = fma(s, mu[j], quad);
        lin = fma(mu[j], q[j], lin);
    }
    out[p] = quad + lin + q0_[p] + trc;   // :132
}

static int gps_layout_ok(int layout, int64_t NP) {
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MINOR && NP != 1) return TRAJOPT_ERR_UNSUPPORTED;
    return TRAJOPT_OK;
}

// device staging for the host-pointer wrappers: one allocation, inputs uploaded, outputs zeroed
struct HostStage {
    double* buf = nullptr;
    size_t used = 0, cap = 0;
    hipStream_t st = nullptr;
    std::vector<std::pair<double*, std::pair<double*, size_t>>> outs;
    ~HostStage() { if (buf) (void)hipFree(buf); }
    int init(size_t doubles) {
        cap = doubles;
        TRAJOPT_HIP_CHECK(hipMalloc(&buf, cap * sizeof(double)));
        TRAJOPT_HIP_CHECK(hipMemsetAsync(buf, 0, cap * sizeof(double), st));
        return TRAJOPT_OK;
    }
    int up(const double* h, size_t cnt, const double** d) {
        if (used + cnt > cap) return TRAJOPT_ERR_BAD_ARG;
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(buf + used, h, cnt * sizeof(double), hipMemcpyHostToDevice, st));
        *d = buf + used;
        used += cnt;
        return TRAJOPT_OK;
    }
    double* out(double* h, size_t cnt) {
        double* d = buf + used;
        used += cnt;
        outs.push_back({h, {d, cnt}});
        return d;
    }
    int down() {
        for (auto& o : outs)
            TRAJOPT_HIP_CHECK(hipMemcpyAsync(o.first, o.second.first, o.second.second * sizeof(double),
                                             hipMemcpyDeviceToHost, st));
        TRAJOPT_HIP_CHECK(hipStreamSynchronize(st));
        return TRAJOPT_OK;
    }
};

struct LaunchFwd {
    const GpsFwdArgs& a; dim3 grid; hipStream_t st;
    template <int N, int M> void operator()() const { hipLaunchKernelGGL((gps_forward_kernel<N, M>), grid, dim3(64), 0, st, a); }
};
struct LaunchAug {
    const GpsAugArgs& a; dim3 grid; hipStream_t st;
    template <int N, int M> void operator()() const { hipLaunchKernelGGL((gps_augment_kernel<N, M>), grid, dim3(64), 0, st, a); }
};
struct LaunchKl {
    const GpsKlArgs& a; dim3 grid; hipStream_t st;
    template <int N, int M> void operator()() const { hipLaunchKernelGGL((gps_kl_kernel<N, M>), grid, dim3(64), 0, st, a); }
};

#define GPS_TRY(expr)                       \
    do {                                    \
        const int rc_ = (expr);             \
        if (rc_ != TRAJOPT_OK) return rc_;  \
    } while (0)

}  // namespace trajopt

using namespace trajopt;

extern "C" {

int trajopt_gps_forward_pass_batched(const double* mu_x0, const double* sigma_x0, const double* A, const double* B,
                                     const double* c, const double* sigma_dyn, const double* K, const double* kff,
                                     const double* sigma_ctl, int n, int m, int T, int64_t NP, int layout,
                                     double* mu_x, double* sigma_x, double* mu_u, double* sigma_u, double* mu_xu,
                                     double* sigma_xu, void* stream) {
    if (!mu_x0 || !sigma_x0 || !A || !B || !c || !sigma_dyn || !K || !kff || !sigma_ctl || !mu_x || !sigma_x || !mu_u ||
        !sigma_u || !mu_xu || !sigma_xu || n <= 0 || m <= 0 || T <= 0 || NP <= 0)
        return TRAJOPT_ERR_BAD_ARG;
    GPS_TRY(gps_layout_ok(layout, NP));
    GpsFwdArgs a{mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, mu_x, sigma_x, mu_u, sigma_u, mu_xu, sigma_xu,
                 (size_t)NP, T};
    const dim3 grid((unsigned)((NP + 63) / 64)), block(64);
    hipStream_t st = as_stream(stream);
    if (!gps_dispatch(n, m, LaunchFwd{a, grid, st}))   // any other dx <= 32, du <= 16: gen_wave.hip
        return launch_gps_forward_generic(mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, n, m, T, NP, mu_x, sigma_x,
                                          mu_u, sigma_u, mu_xu, sigma_xu, st);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

int trajopt_gps_augment_cost_batched(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                     const double* Cxu, const double* c0, const double* K, const double* kff,
                                     const double* sigma_ctl, const double* alpha, int n, int m, int T, int64_t NP,
                                     int layout, double* agCxx, double* agcx, double* agCuu, double* agcu,
                                     double* agCxu, double* agc0, int32_t* status, void* stream) {
    if (!Cxx || !cx || !Cuu || !cu || !Cxu || !c0 || !K || !kff || !sigma_ctl || !alpha || !agCxx || !agcx || !agCuu ||
        !agcu || !agCxu || !agc0 || !status || n <= 0 || m <= 0 || T <= 0 || NP <= 0 || T >= 65535)
        return TRAJOPT_ERR_BAD_ARG;
    GPS_TRY(gps_layout_ok(layout, NP));
    GpsAugArgs a{Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, agCxx, agcx, agCuu, agcu, agCxu, agc0, status,
                 (size_t)NP, T};
    const dim3 grid((unsigned)((NP + 63) / 64), (unsigned)(T + 1)), block(64);
    hipStream_t st = as_stream(stream);
    TRAJOPT_HIP_CHECK(hipMemsetAsync(status, 0, (size_t)NP * sizeof(int32_t), st));
    if (!gps_dispatch(n, m, LaunchAug{a, grid, st}))
        return launch_gps_augment_generic(Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, n, m, T, NP, agCxx, agcx,
                                          agCuu, agcu, agCxu, agc0, status, st);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

int trajopt_gps_kl_divergence_batched(const double* p_K, const double* p_kff, const double* p_sigma_ctl,
                                      const double* q_K, const double* q_kff, const double* q_sigma_ctl,
                                      const double* mu_x, const double* sigma_x, int n, int m, int T, int64_t NP,
                                      int layout, double* kl, int32_t* status, void* stream) {
    if (!p_K || !p_kff || !p_sigma_ctl || !q_K || !q_kff || !q_sigma_ctl || !mu_x || !sigma_x || !kl || !status ||
        n <= 0 || m <= 0 || T <= 0 || NP <= 0 || T > 65535)
        return TRAJOPT_ERR_BAD_ARG;
    GPS_TRY(gps_layout_ok(layout, NP));
    GpsKlArgs a{p_K, p_kff, p_sigma_ctl, q_K, q_kff, q_sigma_ctl, mu_x, sigma_x, kl, status, (size_t)NP, T};
    const dim3 grid((unsigned)((NP + 63) / 64), (unsigned)T), block(64);
    hipStream_t st = as_stream(stream);
    TRAJOPT_HIP_CHECK(hipMemsetAsync(status, 0, (size_t)NP * sizeof(int32_t), st));
    if (!gps_dispatch(n, m, LaunchKl{a, grid, st}))
        return launch_gps_kl_generic(p_K, p_kff, p_sigma_ctl, q_K, q_kff, q_sigma_ctl, mu_x, sigma_x, n, m, T, NP, kl,
                                     status, st);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

int trajopt_gps_quad_expectation_batched(const double* mu, const double* sigma_s, const double* Q, const double* q,
                                         const double* q0, int n, int64_t NP, int layout, double* out, void* stream) {
    if (!mu || !sigma_s || !Q || !q || !q0 || !out || n <= 0 || NP <= 0) return TRAJOPT_ERR_BAD_ARG;
    GPS_TRY(gps_layout_ok(layout, NP));
    const dim3 grid((unsigned)((NP + 63) / 64)), block(64);
    hipStream_t st = as_stream(stream);
    switch (n) {
        case 2: hipLaunchKernelGGL((gps_quadexp_kernel<2>), grid, block, 0, st, mu, sigma_s, Q, q, q0, out, (size_t)NP); break;
        case 3: hipLaunchKernelGGL((gps_quadexp_kernel<3>), grid, block, 0, st, mu, sigma_s, Q, q, q0, out, (size_t)NP); break;
        case 4: hipLaunchKernelGGL((gps_quadexp_kernel<4>), grid, block, 0, st, mu, sigma_s, Q, q, q0, out, (size_t)NP); break;
        case 6: hipLaunchKernelGGL((gps_quadexp_kernel<6>), grid, block, 0, st, mu, sigma_s, Q, q, q0, out, (size_t)NP); break;
        default: return launch_gps_quadexp_generic(mu, sigma_s, Q, q, q0, out, n, NP, st);
    }
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

// ---- host-pointer wrappers with the shapes of the reference's pybind11 entries (one problem) ----------------------

int trajopt_gps_forward_pass_host(const double* mu_x0, const double* sigma_x0, const double* A, const double* B,
                                  const double* c, const double* sigma_dyn, const double* K, const double* kff,
                                  const double* sigma_ctl, int n, int m, int T, double* mu_x, double* sigma_x,
                                  double* mu_u, double* sigma_u, double* mu_xu, double* sigma_xu) {
    if (!mu_x0 || !sigma_x0 || !A || !B || !c || !sigma_dyn || !K || !kff || !sigma_ctl || !mu_x || !sigma_x || !mu_u ||
        !sigma_u || !mu_xu || !sigma_xu || n <= 0 || m <= 0 || T <= 0)
        return TRAJOPT_ERR_BAD_ARG;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    if (!gps_shape_supported(n, m) && !gps_generic_shape_ok(n, m)) return TRAJOPT_ERR_UNSUPPORTED;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m, w = n + m, T1 = (size_t)T + 1;
    HostStage s;
    GPS_TRY(s.init(n + nn + (2 * nn + 2 * nm + n + m + mm) * T + (n + nn + w + w * w) * T1 + (m + mm) * T));
    const double *d[9];
    const double* h[9] = {mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl};
    const size_t cnt[9] = {(size_t)n, nn, nn * T, nm * T, (size_t)n * T, nn * T, nm * T, (size_t)m * T, mm * T};
    for (int i = 0; i < 9; ++i) GPS_TRY(s.up(h[i], cnt[i], &d[i]));
    double *o0 = s.out(mu_x, n * T1), *o1 = s.out(sigma_x, nn * T1), *o2 = s.out(mu_u, (size_t)m * T),
           *o3 = s.out(sigma_u, mm * T), *o4 = s.out(mu_xu, w * T1), *o5 = s.out(sigma_xu, w * w * T1);
    GPS_TRY(trajopt_gps_forward_pass_batched(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], n, m, T, 1,
                                             TRAJOPT_LAYOUT_PROBLEM_MAJOR, o0, o1, o2, o3, o4, o5, s.st));
    return s.down();
}

int trajopt_gps_augment_cost_host(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                  const double* Cxu, const double* c0, const double* K, const double* kff,
                                  const double* sigma_ctl, const double* alpha, int n, int m, int T, double* agCxx,
                                  double* agcx, double* agCuu, double* agcu, double* agCxu, double* agc0, int* status) {
    if (!Cxx || !cx || !Cuu || !cu || !Cxu || !c0 || !K || !kff || !sigma_ctl || !alpha || !agCxx || !agcx || !agCuu ||
        !agcu || !agCxu || !agc0 || !status || n <= 0 || m <= 0 || T <= 0)
        return TRAJOPT_ERR_BAD_ARG;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    if (!gps_shape_supported(n, m) && !gps_generic_shape_ok(n, m)) return TRAJOPT_ERR_UNSUPPORTED;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m, T1 = (size_t)T + 1;
    const size_t per = nn + n + mm + m + nm + 1;
    HostStage s;
    GPS_TRY(s.init(2 * per * T1 + (nm + m + mm + 1) * T + 1));
    const double* d[10];
    const double* h[10] = {Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha};
    const size_t cnt[10] = {nn * T1, n * T1, mm * T1, m * T1, nm * T1, T1, nm * T, (size_t)m * T, mm * T, (size_t)T};
    for (int i = 0; i < 10; ++i) GPS_TRY(s.up(h[i], cnt[i], &d[i]));
    double *o0 = s.out(agCxx, nn * T1), *o1 = s.out(agcx, n * T1), *o2 = s.out(agCuu, mm * T1),
           *o3 = s.out(agcu, m * T1), *o4 = s.out(agCxu, nm * T1), *o5 = s.out(agc0, T1);
    int32_t* d_status = reinterpret_cast<int32_t*>(s.buf + s.used);
    GPS_TRY(trajopt_gps_augment_cost_batched(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], n, m, T, 1,
                                             TRAJOPT_LAYOUT_PROBLEM_MAJOR, o0, o1, o2, o3, o4, o5, d_status, s.st));
    int32_t stt = 0;
    TRAJOPT_HIP_CHECK(hipMemcpyAsync(&stt, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s.st));
    GPS_TRY(s.down());
    *status = stt;
    return TRAJOPT_OK;
}

int trajopt_gps_kl_divergence_host(const double* p_K, const double* p_kff, const double* p_sigma_ctl,
                                   const double* q_K, const double* q_kff, const double* q_sigma_ctl,
                                   const double* mu_x, const double* sigma_x, int n, int m, int T, double* kl,
                                   int* status) {
    if (!p_K || !p_kff || !p_sigma_ctl || !q_K || !q_kff || !q_sigma_ctl || !mu_x || !sigma_x || !kl || !status ||
        n <= 0 || m <= 0 || T <= 0)
        return TRAJOPT_ERR_BAD_ARG;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    if (!gps_shape_supported(n, m) && !gps_generic_shape_ok(n, m)) return TRAJOPT_ERR_UNSUPPORTED;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    HostStage s;
    // mu_x / sigma_x may hold T + 1 slices (mbgps.py:198-201 passes xdist.mu, xdist.sigma); the first T are read
    GPS_TRY(s.init((2 * (nm + m + mm) + n + nn + 1) * (size_t)T + 1));
    const double* d[8];
    const double* h[8] = {p_K, p_kff, p_sigma_ctl, q_K, q_kff, q_sigma_ctl, mu_x, sigma_x};
    const size_t cnt[8] = {nm * T, (size_t)m * T, mm * T, nm * T, (size_t)m * T, mm * T, (size_t)n * T, nn * T};
    for (int i = 0; i < 8; ++i) GPS_TRY(s.up(h[i], cnt[i], &d[i]));
    double* o = s.out(kl, (size_t)T);
    int32_t* d_status = reinterpret_cast<int32_t*>(s.buf + s.used);
    GPS_TRY(trajopt_gps_kl_divergence_batched(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], n, m, T, 1,
                                              TRAJOPT_LAYOUT_PROBLEM_MAJOR, o, d_status, s.st));
    int32_t stt = 0;
    TRAJOPT_HIP_CHECK(hipMemcpyAsync(&stt, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s.st));
    GPS_TRY(s.down());
    *status = stt;
    return TRAJOPT_OK;
}

int trajopt_gps_quad_expectation_host(const double* mu, const double* sigma_s, const double* Q, const double* q,
                                      double q0, int n, double* result) {
    if (!mu || !sigma_s || !Q || !q || !result || n <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    if (n > 64) return TRAJOPT_ERR_UNSUPPORTED;
    const size_t nn = (size_t)n * n;
    HostStage s;
    GPS_TRY(s.init(2 * nn + 2 * n + 2));
    const double* d[5];
    const double* h[5] = {mu, sigma_s, Q, q, &q0};
    const size_t cnt[5] = {(size_t)n, nn, nn, (size_t)n, 1};
    for (int i = 0; i < 5; ++i) GPS_TRY(s.up(h[i], cnt[i], &d[i]));
    double* o = s.out(result, 1);
    GPS_TRY(trajopt_gps_quad_expectation_batched(d[0], d[1], d[2], d[3], d[4], n, 1, TRAJOPT_LAYOUT_PROBLEM_MAJOR, o, s.st));
    return s.down();
}

}  // extern "C"
