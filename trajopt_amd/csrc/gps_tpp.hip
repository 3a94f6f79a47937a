// Max-entropy (GPS) backward pass, lane = problem (the first "next" row after the iLQR path).
//
//   /root/reference/trajopt/gps/src/util.cpp:272-374   backward_pass(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha)
//
// The recursion skeleton of bwd_tpp.hip with the affine term c, the process noise sigma_dyn, the constant c0 and a
// per-step temperature alpha:  Q.. = -(C.. + F'V'F) / alpha,  K = -Quu^-1 Qux,  kff = -1/2 Quu^-1 qu,
// sigma_ctl = sym(-1/2 Quu^-1),  V = sym(-alpha (Qxx + Qux'K)),  v = -alpha (qx + 2 Qux'kff),
// v0 = -alpha (1/2 qu'kff + q0 + 1/2 (m log 2pi - log det(-2 Quu))).
// Mapping and addressing as in bwd_tpp.hip: PROBLEM_MINOR, every access of a wave one coalesced row, V / v and the
// step temporaries in VGPRs, the next step's inputs loaded (unconditionally, index clamped) before this step's
// arithmetic.  MBGPS calls this pass dozens of times per dual solve (gps/mbgps.py:135-201) with small systems, so the
// shapes instantiated are the reference's own: du <= 2.
#include "gps_common.h"

namespace trajopt {

struct GpsArgs {
    trajopt_gps_backward_io_t io;
    size_t NP;
    int T;
};

template <int N, int M>
struct GpsIn {
    double A[N * N], B[N * M], c[N], Cuu[M * M], Cxu[N * M], cx[N], cu[M], c0, alpha;
    double S[N <= 4 ? N * N : 1], Cxx[N <= 4 ? N * N : 1];   // prefetched with the rest up to dx = 4 (see below)
};
// At dx = 6 sigma_dyn[t] and Cxx[t] are not part of the prefetched set: the step loads them at its top and first uses
// them after the V'F products, which keeps two sets + V + the step temporaries inside 512 VGPRs.

template <int N, int M>
__device__ __forceinline__ void gps_load(GpsIn<N, M>& s, const trajopt_gps_backward_io_t& io, int t, size_t NP, size_t p) {
    gld<N * N>(s.A, io.A, N * N, t, NP, p);
    gld<N * M>(s.B, io.B, N * M, t, NP, p);
    gld<N>(s.c, io.c, N, t, NP, p);
    gld<M * M>(s.Cuu, io.Cuu, M * M, t, NP, p);
    gld<N * M>(s.Cxu, io.Cxu, N * M, t, NP, p);
    gld<N>(s.cx, io.cx, N, t, NP, p);
    gld<M>(s.cu, io.cu, M, t, NP, p);
    s.c0 = io.c0[(size_t)t * NP + p];
    s.alpha = io.alpha[(size_t)t * NP + p];
    if constexpr (N <= 4) {
        gld<N * N>(s.S, io.sigma_dyn, N * N, t, NP, p);
        gld<N * N>(s.Cxx, io.Cxx, N * N, t, NP, p);
    }
}

template <int N, int M, bool FULL>
__global__ __launch_bounds__(64) void gps_tpp_kernel(GpsArgs a) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x, NP = a.NP;
    if (p >= NP) return;
    const trajopt_gps_backward_io_t& io = a.io;
    const int T = a.T;
    double V[N * N], v[N], v0;
    gld<N * N>(V, io.Cxx, N * N, T, NP, p);  // util.cpp:318-320
    gld<N>(v, io.cx, N, T, NP, p);
    v0 = io.c0[(size_t)T * NP + p];
    gst<N * N>(io.V, N * N, T, NP, p, V);
    gst<N>(io.v, N, T, NP, p, v);
    io.v0[(size_t)T * NP + p] = v0;
    int status = 0, where = -1;

    auto step = [&](const GpsIn<N, M>& in, int t) -> bool {
        const double na = -1.0 / in.alpha;
        double W[N * N], Z[N * M], Vc[N], Sg[N * N], Cx[N * N];
        if constexpr (N <= 4) {
#pragma unroll
            for (int k = 0; k < N * N; ++k) { Sg[k] = in.S[k]; Cx[k] = in.Cxx[k]; }
        } else {
            gld<N * N>(Sg, io.sigma_dyn, N * N, t, NP, p);
            gld<N * N>(Cx, io.Cxx, N * N, t, NP, p);
        }
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) s = fma(V[i + N * k], in.A[k + N * j], s);
                W[i + N * j] = s;
            }
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) s = fma(V[i + N * k], in.B[k + N * j], s);
                Z[i + N * j] = s;
            }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(V[i + N * k], in.c[k], s);
            Vc[i] = s;
        }
        double Qxx[N * N], Qxu[N * M], Quu[M * M], qx[N], qu[M];
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = Cx[i + N * j];
#pragma unroll
                for (int k = 0; k < N; ++k) s = fma(in.A[k + N * i], W[k + N * j], s);
                Qxx[i + N * j] = na * s;  // :324
            }
#pragma unroll
        for (int j = 0; j < M; ++j) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = in.Cxu[i + N * j];
#pragma unroll
                for (int k = 0; k < N; ++k) s = fma(in.A[k + N * i], Z[k + N * j], s);
                Qxu[i + N * j] = na * s;  // Qux(j, i) = Qxu(i, j), :326
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double s = in.Cuu[i + M * j];
#pragma unroll
                for (int k = 0; k < N; ++k) s = fma(in.B[k + N * i], Z[k + N * j], s);
                Quu[i + M * j] = na * s;  // :325
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = in.cx[i];
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(in.A[k + N * i], fma(2.0, Vc[k], v[k]), s);
            qx[i] = na * s;  // :329
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double s = in.cu[i];
#pragma unroll
            for (int k = 0; k < N; ++k) s = fma(in.B[k + N * i], fma(2.0, Vc[k], v[k]), s);
            qu[i] = na * s;  // :328
        }
        double s0 = in.c0 + v0;  // :330-331
#pragma unroll
        for (int i = 0; i < N; ++i) {
            s0 = fma(in.c[i], Vc[i] + v[i], s0);
#pragma unroll
            for (int k = 0; k < N; ++k) s0 = fma(V[i + N * k], Sg[k + N * i], s0);  // trace(V' sigma_dyn)
        }
        const double q0 = na * s0;
        if constexpr (FULL) {
            if (io.Qxx) gst<N * N>(io.Qxx, N * N, t, NP, p, Qxx);
            if (io.Quu) gst<M * M>(io.Quu, M * M, t, NP, p, Quu);
            if (io.qx) gst<N>(io.qx, N, t, NP, p, qx);
            if (io.qu) gst<M>(io.qu, M, t, NP, p, qu);
            if (io.q0) io.q0[(size_t)t * NP + p] = q0;
            if (io.Qux) {
                double Qux[M * N];
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = 0; j < M; ++j) Qux[j + M * i] = Qxu[i + N * j];
                gst<M * N>(io.Qux, N * M, t, NP, p, Qux);
            }
        }
        {   // util.cpp:333-336: a positive definite Quu (the maximisation needs it negative definite) stops the sweep
            double L[M * M];
            if (sympd_factor<M>(Quu, L)) { status = 1; where = t; return false; }
        }
        double Qi[M * M], det;
        if (!small_inv<M>(Quu, Qi, det)) { status = 2; where = t; return false; }  // inv() throws in the reference
        double K[M * N], kff[M], sg[M * M];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < M; ++k) s = fma(Qi[j + M * k], Qxu[i + N * k], s);
                K[j + M * i] = -s;  // :339
            }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) s = fma(Qi[j + M * k], qu[k], s);
            kff[j] = -0.5 * s;  // :340
        }
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < M; ++i) sg[i + M * j] = -0.25 * (Qi[i + M * j] + Qi[j + M * i]);  // :342-343
        // V = sym(-alpha (Qxx + Qux'K)), v = -alpha (qx + 2 Qux'kff)   (:348-351)
        const double al = in.alpha;
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = Qxx[i + N * j];
#pragma unroll
                for (int k = 0; k < M; ++k) s = fma(Qxu[i + N * k], K[k + M * j], s);
                W[i + N * j] = -al * s;
            }
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) V[i + N * j] = 0.5 * (W[i + N * j] + W[j + N * i]);
        double qk = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = qx[i];
#pragma unroll
            for (int k = 0; k < M; ++k) s = fma(2.0 * Qxu[i + N * k], kff[k], s);
            v[i] = -al * s;
        }
#pragma unroll
        for (int k = 0; k < M; ++k) qk = fma(qu[k], kff[k], qk);
        const double det2 = (M == 1 ? -2.0 : 4.0) * det;  // det(-2 Quu)
        v0 = -al * (0.5 * qk + q0 + 0.5 * (M * 1.8378770664093453 - log(det2)));  // :352-353, log(2 pi)
        gst<M * N>(io.K, N * M, t, NP, p, K);
        gst<M>(io.kff, M, t, NP, p, kff);
        gst<M * M>(io.sigma_ctl, M * M, t, NP, p, sg);
        gst<N * N>(io.V, N * N, t, NP, p, V);
        gst<N>(io.v, N, t, NP, p, v);
        io.v0[(size_t)t * NP + p] = v0;
        return true;
    };

    GpsIn<N, M> s0, s1;
    int t = T - 1;
    gps_load<N, M>(s0, io, t, NP, p);
    while (t >= 0) {
        gps_load<N, M>(s1, io, t >= 1 ? t - 1 : 0, NP, p);
        if (!step(s0, t)) break;
        if (--t < 0) break;
        gps_load<N, M>(s0, io, t >= 1 ? t - 1 : 0, NP, p);
        if (!step(s1, t)) break;
        --t;
    }
    io.diverge[p] = where;
    if (io.status) io.status[p] = status;
}

template <int N, int M>
static int gps_launch(const GpsArgs& a, hipStream_t st) {
    const bool full = a.io.Qxx || a.io.Qux || a.io.Quu || a.io.qx || a.io.qu || a.io.q0;
    const dim3 grid((unsigned)((a.NP + 63) / 64)), block(64);
    if (full) hipLaunchKernelGGL((gps_tpp_kernel<N, M, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gps_tpp_kernel<N, M, false>), grid, block, 0, st, a);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

}  // namespace trajopt

using namespace trajopt;

extern "C" {

int trajopt_gps_backward_supported(int n, int m) { return (gps_shape_supported(n, m) || gps_generic_shape_ok(n, m)) ? 1 : 0; }

int trajopt_gps_backward_pass_batched(const trajopt_gps_backward_io_t* io, int n, int m, int T, int64_t NP, int layout,
                                      void* stream) {
    if (!io || n <= 0 || m <= 0 || T <= 0 || NP <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (!io->Cxx || !io->cx || !io->Cuu || !io->cu || !io->Cxu || !io->c0 || !io->A || !io->B || !io->c ||
        !io->sigma_dyn || !io->alpha || !io->V || !io->v || !io->v0 || !io->K || !io->kff || !io->sigma_ctl || !io->diverge)
        return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MINOR && NP != 1) return TRAJOPT_ERR_UNSUPPORTED;  // (one problem: same memory)
    if (!trajopt_gps_backward_supported(n, m)) return TRAJOPT_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if (!gps_shape_supported(n, m)) return launch_gps_backward_generic(*io, n, m, T, NP, st);   // any other dx <= 32, du <= 16
    GpsArgs a;
    a.io = *io; a.NP = (size_t)NP; a.T = T;
    if (m == 1) {
        if (n == 2) return gps_launch<2, 1>(a, st);
        if (n == 3) return gps_launch<3, 1>(a, st);
        if (n == 4) return gps_launch<4, 1>(a, st);
        return gps_launch<6, 1>(a, st);
    }
    if (n == 2) return gps_launch<2, 2>(a, st);
    return gps_launch<4, 2>(a, st);
}

// Host-pointer wrapper with the shape of the reference's pybind11 entry (gps/src/util.cpp:272-275, :355-371).
// *diverge: the reference's int (0 = none, else the time index whose Quu was positive definite); *status: 0 ok,
// 1 = that PD stop, 2 = singular Quu at *diverge (Armadillo's inv() throws there).
int trajopt_gps_backward_pass_host(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                   const double* Cxu, const double* c0, const double* A, const double* B,
                                   const double* c, const double* sigma_dyn, const double* alpha, int n, int m, int T,
                                   double* Qxx, double* Qux, double* Quu, double* qx, double* qu, double* q0, double* V,
                                   double* v, double* v0, double* K, double* kff, double* sigma_ctl, int* diverge,
                                   int* status) {
    if (!Cxx || !cx || !Cuu || !cu || !Cxu || !c0 || !A || !B || !c || !sigma_dyn || !alpha || !Qxx || !Qux || !Quu ||
        !qx || !qu || !q0 || !V || !v || !v0 || !K || !kff || !sigma_ctl || !diverge)
        return TRAJOPT_ERR_BAD_ARG;
    if (n <= 0 || m <= 0 || T <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    if (!trajopt_gps_backward_supported(n, m)) return TRAJOPT_ERR_UNSUPPORTED;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m, T1 = (size_t)T + 1;
    const size_t in_counts[11] = {nn * T1, n * T1, mm * T1, m * T1, nm * T1, T1, nn * T, nm * T, (size_t)n * T, nn * T, (size_t)T};
    const double* in_host[11] = {Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha};
    const size_t out_counts[12] = {nn * T, nm * T, mm * T, (size_t)n * T, (size_t)m * T, (size_t)T, nn * T1, n * T1, T1, nm * T,
                                   (size_t)m * T, mm * T};
    double* out_host[12] = {Qxx, Qux, Quu, qx, qu, q0, V, v, v0, K, kff, sigma_ctl};
    size_t total = 2, in_total = 0;
    for (size_t x : in_counts) in_total += x;
    total += in_total;
    for (size_t x : out_counts) total += x;
    double* buf = nullptr;
    TRAJOPT_HIP_CHECK(hipMalloc(&buf, total * sizeof(double)));
    struct Free { double* p; ~Free() { (void)hipFree(p); } } guard{buf};
    hipStream_t st = nullptr;
    double* d = buf;
    const double* din[11];
    for (int i = 0; i < 11; ++i) {
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(d, in_host[i], in_counts[i] * sizeof(double), hipMemcpyHostToDevice, st));
        din[i] = d;
        d += in_counts[i];
    }
    TRAJOPT_HIP_CHECK(hipMemsetAsync(d, 0, (total - in_total) * sizeof(double), st));
    double* dout[12];
    for (int i = 0; i < 12; ++i) { dout[i] = d; d += out_counts[i]; }
    int32_t* d_div = reinterpret_cast<int32_t*>(d);
    int32_t* d_status = reinterpret_cast<int32_t*>(d + 1);
    trajopt_gps_backward_io_t io{};
    io.Cxx = din[0]; io.cx = din[1]; io.Cuu = din[2]; io.cu = din[3]; io.Cxu = din[4]; io.c0 = din[5]; io.A = din[6];
    io.B = din[7]; io.c = din[8]; io.sigma_dyn = din[9]; io.alpha = din[10];
    io.Qxx = dout[0]; io.Qux = dout[1]; io.Quu = dout[2]; io.qx = dout[3]; io.qu = dout[4]; io.q0 = dout[5];
    io.V = dout[6]; io.v = dout[7]; io.v0 = dout[8]; io.K = dout[9]; io.kff = dout[10]; io.sigma_ctl = dout[11];
    io.diverge = d_div; io.status = d_status;
    const int rc = trajopt_gps_backward_pass_batched(&io, n, m, T, 1, TRAJOPT_LAYOUT_PROBLEM_MAJOR, st);
    if (rc != TRAJOPT_OK) return rc;
    for (int i = 0; i < 12; ++i)
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(out_host[i], dout[i], out_counts[i] * sizeof(double), hipMemcpyDeviceToHost, st));
    int32_t div = -1, stt = 0;
    TRAJOPT_HIP_CHECK(hipMemcpyAsync(&div, d_div, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TRAJOPT_HIP_CHECK(hipMemcpyAsync(&stt, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TRAJOPT_HIP_CHECK(hipStreamSynchronize(st));
    *diverge = (stt == 1 && div > 0) ? div : (stt == 2 ? div : 0);  // util.cpp:316,334
    if (status) *status = stt;
    return TRAJOPT_OK;
}

}  // extern "C"
