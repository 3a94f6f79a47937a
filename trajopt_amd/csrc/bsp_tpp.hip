// Belief-space iLQR backward pass, lane = problem:
//   /root/reference/trajopt/bspilqr/src/util.cpp:83-204   backward_pass(Q, q, R, r, P, p, F, G, T, U, V, X, Y, Z, lmbda, reg, ...)
// The iLQR recursion on the belief mean (dimension b) with the covariance channel tau (dimension b^2) riding along:
//   c = q + F's + T'tau + 1/2 X'vec(S),  d = r + G's + V'tau + 1/2 Z'vec(S),  e = p + U'tau + 1/2 Y'vec(S),  tau <- e.
// PROBLEM_MINOR, every access of a wave one coalesced row.  S, s, tau stay in VGPRs; the wide operands (U, Y are
// b^2 x b^2: 256 doubles each per step at b = 4) are streamed straight from HBM into their dot products instead of
// being held: this pass is bound by those bytes (8 (2b^4 + 2b^3 + 2b^2 m + ...) per step, 6.3 KB at b = 4, m = 2).
// Reference quirks (oracle/bsp_oracle.hpp): dS starts at zero here; tau_mode picks what util.cpp:158's read of the
// not-yet-assigned tau.col(i) means (0: tau.col(i+1), the evident intent; 1: zeros).
#include "gps_common.h"

namespace trajopt {

struct BspArgs {
    trajopt_bsp_backward_io_t io;
    size_t NP;
    int T, reg, tau_mode;
};

template <int B, int M>
__global__ __launch_bounds__(64) void bsp_tpp_kernel(BspArgs a) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x, NP = a.NP;
    if (p >= NP) return;
    constexpr int BB = B * B;
    const trajopt_bsp_backward_io_t& io = a.io;
    const int T = a.T, reg = a.reg;
    const double lam = io.lmbda[p];
    double S[BB], s[B], tau[BB];
    gld<BB>(S, io.Q, BB, T, NP, p);   // :142-144
    gld<B>(s, io.q, B, T, NP, p);
    gld<BB>(tau, io.p, BB, T, NP, p);
    gst<BB>(io.S, BB, T, NP, p, S);
    gst<B>(io.s, B, T, NP, p, s);
    gst<BB>(io.tau, BB, T, NP, p, tau);
    double dS0 = 0.0, dS1 = 0.0;
    int where = -1;

    // b = 4: the covariance channel's operands are streamed, two columns (2 b^2 = 32 loads, issued back to back into
    // registers) at a time, by RUN-TIME loops over the columns with walking pointers; the vectors they multiply
    // (vec(S), tau) and the sums they produce sit in this lane's column of LDS, which takes a run-time index where a
    // register array cannot.  Fully unrolled, the 800 loads of a step each carried a loop-invariant 64-bit offset that
    // hipcc hoisted out of the time loop (830 spilled registers); one column per iteration with the loads feeding the
    // multiply-adds directly ran the loads one after another (0.3 us each): both 26 ms where this form takes 1.9.
    __shared__ double l_S[BB][64], l_tau[BB][64], l_e[BB][64], l_c[B][64], l_d[M][64];
    const int ln = threadIdx.x;
    // sum_k Mat[k, j] v[k] for columns j of a (BB x cols) operand at slice t -> out[j][ln] (+= when ACC)
    auto stream = [&](const double* arr, int cols, int t, const double (*v)[64], double (*out)[64], double scale,
                      bool acc) {
        const double* col = arr + ((size_t)t * BB * cols) * NP + p;
        // JW columns (JW b^2 loads) in flight at a time: a lone wave's loads each pay a full miss (the rows of a
        // step lie NP * 8 bytes apart), so the count in flight is what sets the pace
        constexpr int JW = 2;
        int j = 0;
#pragma unroll 1
        for (; j + JW <= cols; j += JW, col += (size_t)JW * BB * NP) {
            double x[JW][BB];
#pragma unroll
            for (int jj = 0; jj < JW; ++jj)
#pragma unroll
                for (int k = 0; k < BB; ++k) x[jj][k] = col[((size_t)jj * BB + k) * NP];
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) {
                double sacc = 0.0;
#pragma unroll
                for (int k = 0; k < BB; ++k) sacc = fma(x[jj][k], v[k][ln], sacc);
                out[j + jj][ln] = acc ? fma(scale, sacc, out[j + jj][ln]) : scale * sacc;
            }
        }
#pragma unroll 1
        for (; j < cols; ++j, col += (size_t)BB * NP) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < BB; ++k) sacc = fma(col[(size_t)k * NP], v[k][ln], sacc);
            out[j][ln] = acc ? fma(scale, sacc, out[j][ln]) : scale * sacc;
        }
    };

    for (int t = T - 1; t >= 0; --t) {
        double F[BB], G[B * M], R[M * M], P[B * M];
        gld<BB>(F, io.F, BB, t, NP, p);
        gld<B * M>(G, io.G, B * M, t, NP, p);
        gld<M * M>(R, io.R, M * M, t, NP, p);
        gld<B * M>(P, io.P, B * M, t, NP, p);
        double c[B], d[M], e[BB];
        gld<B>(c, io.q, B, t, NP, p);
        gld<M>(d, io.r, M, t, NP, p);
        gld<BB>(e, io.p, BB, t, NP, p);
        // the covariance channel (:152-158)
        if constexpr (BB <= 4) {
            // b = 2: the six operands are 64 doubles in all - load them in one go, multiply from registers
            double mT[BB * B], mX[BB * B], mV[BB * M], mZ[BB * M], mU[BB * BB], mY[BB * BB];
            gld<BB * B>(mT, io.T, BB * B, t, NP, p);
            gld<BB * B>(mX, io.X, BB * B, t, NP, p);
            gld<BB * M>(mV, io.V, BB * M, t, NP, p);
            gld<BB * M>(mZ, io.Z, BB * M, t, NP, p);
            gld<BB * BB>(mY, io.Y, BB * BB, t, NP, p);
            if (a.tau_mode == 0) gld<BB * BB>(mU, io.U, BB * BB, t, NP, p);
            auto dot = [&](const double* mat, int j, const double (&v)[BB]) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < BB; ++k) x = fma(mat[k + BB * j], v[k], x);
                return x;
            };
#pragma unroll
            for (int j = 0; j < B; ++j) c[j] = fma(0.5, dot(mX, j, S), c[j] + dot(mT, j, tau));
#pragma unroll
            for (int j = 0; j < M; ++j) d[j] = fma(0.5, dot(mZ, j, S), d[j] + dot(mV, j, tau));
#pragma unroll
            for (int j = 0; j < BB; ++j) {
                if (a.tau_mode == 0) e[j] += dot(mU, j, tau);
                e[j] = fma(0.5, dot(mY, j, S), e[j]);
            }
        } else {
            // b = 4: streamed operands
#pragma unroll
            for (int k = 0; k < BB; ++k) { l_S[k][ln] = S[k]; l_tau[k][ln] = tau[k]; }   // vectorise(S) = the array itself
            stream(io.T, B, t, l_tau, l_c, 1.0, false);
            stream(io.X, B, t, l_S, l_c, 0.5, true);
            stream(io.V, M, t, l_tau, l_d, 1.0, false);
            stream(io.Z, M, t, l_S, l_d, 0.5, true);
            stream(io.Y, BB, t, l_S, l_e, 0.5, false);
            if (a.tau_mode == 0) stream(io.U, BB, t, l_tau, l_e, 1.0, true);
#pragma unroll
            for (int i = 0; i < B; ++i) c[i] += l_c[i][ln];
#pragma unroll
            for (int i = 0; i < M; ++i) d[i] += l_d[i][ln];
#pragma unroll
            for (int k = 0; k < BB; ++k) e[k] += l_e[k][ln];
        }
        // W = S F, Z = S G
        double W[BB], Zg[B * M];
#pragma unroll
        for (int j = 0; j < B; ++j)
#pragma unroll
            for (int i = 0; i < B; ++i) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < B; ++k) x = fma(S[i + B * k], F[k + B * j], x);
                W[i + B * j] = x;
            }
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < B; ++i) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < B; ++k) x = fma(S[i + B * k], G[k + B * j], x);
                Zg[i + B * j] = x;
            }
        double C[BB], D[M * M], Et[B * M];   // Et(i, j) = E(j, i) = (P + F'SG)(i, j)
        gld<BB>(C, io.Q, BB, t, NP, p);
#pragma unroll
        for (int j = 0; j < B; ++j)
#pragma unroll
            for (int i = 0; i < B; ++i) {
                double x = C[i + B * j];
#pragma unroll
                for (int k = 0; k < B; ++k) x = fma(F[k + B * i], W[k + B * j], x);
                C[i + B * j] = x;   // :148
            }
#pragma unroll
        for (int j = 0; j < M; ++j) {
#pragma unroll
            for (int i = 0; i < B; ++i) {
                double x = P[i + B * j];
#pragma unroll
                for (int k = 0; k < B; ++k) x = fma(F[k + B * i], Zg[k + B * j], x);
                Et[i + B * j] = x;  // :150
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double x = R[i + M * j];
#pragma unroll
                for (int k = 0; k < B; ++k) x = fma(G[k + B * i], Zg[k + B * j], x);
                D[i + M * j] = x;   // :149
            }
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            double x = c[i];
#pragma unroll
            for (int k = 0; k < B; ++k) x = fma(F[k + B * i], s[k], x);
            c[i] = x;               // :152
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double x = d[i];
#pragma unroll
            for (int k = 0; k < B; ++k) x = fma(G[k + B * i], s[k], x);
            d[i] = x;               // :155
        }
        // regularised copies (:160-168): Sreg = S + lam I (reg 2) -> F'SregG = F'SG + lam F'G, G'SregG = G'SG + lam G'G
        double Ereg_t[B * M], Dreg[M * M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
#pragma unroll
            for (int i = 0; i < B; ++i) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < B; ++k) x = fma(F[k + B * i], G[k + B * j], x);
                Ereg_t[i + B * j] = reg == 2 ? fma(lam, x, Et[i + B * j]) : Et[i + B * j];
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < B; ++k) x = fma(G[k + B * i], G[k + B * j], x);
                double v = reg == 2 ? fma(lam, x, D[i + M * j]) : D[i + M * j];
                if (reg == 1 && i == j) v += lam;
                Dreg[i + M * j] = v;
            }
        }
        double L[M * M];
        if (!sympd_factor<M>(Dreg, L)) { where = t; break; }   // :170-173
        double Di[M * M], det;
        small_inv<M>(Dreg, Di, det);                           // :175
        double K[M * B], kff[M];
#pragma unroll
        for (int i = 0; i < B; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < M; ++k) x = fma(Di[j + M * k], Ereg_t[i + B * k], x);
                K[j + M * i] = -x;  // :176
            }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double x = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) x = fma(Di[j + M * k], d[k], x);
            kff[j] = -x;            // :177
        }
        double Dk[M], DK[M * B];    // D kff, D K
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double x = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) x = fma(D[i + M * k], kff[k], x);
            Dk[i] = x;
#pragma unroll
            for (int j = 0; j < B; ++j) {
                double y = 0.0;
#pragma unroll
                for (int k = 0; k < M; ++k) y = fma(D[i + M * k], K[k + M * j], y);
                DK[i + M * j] = y;
            }
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {   // :179
            dS0 = fma(kff[j], d[j], dS0);
            dS1 = fma(0.5 * kff[j], Dk[j], dS1);
        }
        // s = c + K'(D kff) + K'd + E'kff;  S = sym(C + K'(D K) + K'E + E'K)   (:183-188)
#pragma unroll
        for (int i = 0; i < B; ++i) {
            double x = c[i];
#pragma unroll
            for (int k = 0; k < M; ++k) x = fma(K[k + M * i], Dk[k] + d[k], fma(Et[i + B * k], kff[k], x));
            s[i] = x;
        }
#pragma unroll
        for (int j = 0; j < B; ++j)
#pragma unroll
            for (int i = 0; i < B; ++i) {
                double x = C[i + B * j];
#pragma unroll
                for (int k = 0; k < M; ++k)
                    x = fma(K[k + M * i], DK[k + M * j] + Et[j + B * k], fma(Et[i + B * k], K[k + M * j], x));
                W[i + B * j] = x;
            }
#pragma unroll
        for (int j = 0; j < B; ++j)
#pragma unroll
            for (int i = 0; i < B; ++i) S[i + B * j] = 0.5 * (W[i + B * j] + W[j + B * i]);
#pragma unroll
        for (int k = 0; k < BB; ++k) tau[k] = e[k];   // :181
        gst<M * B>(io.K, B * M, t, NP, p, K);
        gst<M>(io.kff, M, t, NP, p, kff);
        gst<BB>(io.S, BB, t, NP, p, S);
        gst<B>(io.s, B, t, NP, p, s);
        gst<BB>(io.tau, BB, t, NP, p, tau);
    }
    io.dS[p] = dS0;
    io.dS[NP + p] = dS1;
    io.diverge[p] = where;
}

struct LaunchBsp {
    const BspArgs& a; dim3 grid; hipStream_t st;
    template <int B, int M> void operator()() const { hipLaunchKernelGGL((bsp_tpp_kernel<B, M>), grid, dim3(64), 0, st, a); }
};

template <class F>
inline bool bsp_dispatch(int b, int m, F&& fn) {
    if (b == 2 && m == 1) { fn.template operator()<2, 1>(); return true; }
    if (b == 2 && m == 2) { fn.template operator()<2, 2>(); return true; }
    if (b == 4 && m == 1) { fn.template operator()<4, 1>(); return true; }
    if (b == 4 && m == 2) { fn.template operator()<4, 2>(); return true; }
    return false;
}

}  // namespace trajopt

using namespace trajopt;

extern "C" {

static bool bsp_lane_shape(int b, int m) { return (b == 2 || b == 4) && (m == 1 || m == 2); }
int trajopt_bspilqr_backward_supported(int b, int m) { return (bsp_lane_shape(b, m) || bsp_generic_shape_ok(b, m)) ? 1 : 0; }

int trajopt_bspilqr_backward_pass_batched(const trajopt_bsp_backward_io_t* io, int reg, int tau_mode, int b, int m,
                                          int T, int64_t NP, int layout, void* stream) {
    if (!io || b <= 0 || m <= 0 || T <= 0 || NP <= 0 || (reg != 1 && reg != 2) || (tau_mode != 0 && tau_mode != 1))
        return TRAJOPT_ERR_BAD_ARG;
    if (!io->Q || !io->q || !io->R || !io->r || !io->P || !io->p || !io->F || !io->G || !io->T || !io->U || !io->V ||
        !io->X || !io->Y || !io->Z || !io->lmbda || !io->S || !io->s || !io->tau || !io->dS || !io->K || !io->kff ||
        !io->diverge)
        return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MINOR && NP != 1) return TRAJOPT_ERR_UNSUPPORTED;
    BspArgs a;
    a.io = *io; a.NP = (size_t)NP; a.T = T; a.reg = reg; a.tau_mode = tau_mode;
    const dim3 grid((unsigned)((NP + 63) / 64));
    hipStream_t st = as_stream(stream);
    if (!bsp_lane_shape(b, m)) return launch_bsp_backward_generic(*io, reg, tau_mode, b, m, T, NP, st);   // any other b <= 8, du <= 8
    if (!bsp_dispatch(b, m, LaunchBsp{a, grid, st})) return TRAJOPT_ERR_UNSUPPORTED;
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

// Host-pointer wrapper with the shape of the reference's pybind11 entry (bspilqr/src/util.cpp:83-91, :201-202).
// Q, q, p hold T + 1 slices, everything else T.  *diverge as the reference's int (0 = none); *failed disambiguates.
int trajopt_bspilqr_backward_pass_host(const double* Q, const double* q, const double* R, const double* r,
                                       const double* P, const double* p, const double* F, const double* G,
                                       const double* T_, const double* U, const double* V, const double* X,
                                       const double* Y, const double* Z, double lmbda, int reg, int tau_mode, int b,
                                       int m, int T, double* S, double* s, double* tau, double* dS, double* K,
                                       double* kff, int* diverge, int* failed) {
    if (!Q || !q || !R || !r || !P || !p || !F || !G || !T_ || !U || !V || !X || !Y || !Z || !S || !s || !tau || !dS ||
        !K || !kff || !diverge || b <= 0 || m <= 0 || T <= 0)
        return TRAJOPT_ERR_BAD_ARG;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    if (!trajopt_bspilqr_backward_supported(b, m)) return TRAJOPT_ERR_UNSUPPORTED;
    const size_t bb = (size_t)b * b, bm = (size_t)b * m, mm = (size_t)m * m, T1 = (size_t)T + 1, Ts = (size_t)T;
    const double* in_host[14] = {Q, q, R, r, P, p, F, G, T_, U, V, X, Y, Z};
    const size_t in_cnt[14] = {bb * T1, b * T1, mm * Ts, m * Ts, bm * Ts, bb * T1, bb * Ts, bm * Ts, bb * b * Ts,
                               bb * bb * Ts, bb * m * Ts, bb * b * Ts, bb * bb * Ts, bb * m * Ts};
    double* out_host[6] = {S, s, tau, dS, K, kff};
    const size_t out_cnt[6] = {bb * T1, b * T1, bb * T1, 2, bm * Ts, m * Ts};
    size_t in_total = 0, out_total = 0;
    for (size_t x : in_cnt) in_total += x;
    for (size_t x : out_cnt) out_total += x;
    double* buf = nullptr;
    TRAJOPT_HIP_CHECK(hipMalloc(&buf, (in_total + out_total + 2) * sizeof(double)));
    struct Free { double* p; ~Free() { (void)hipFree(p); } } guard{buf};
    hipStream_t st = nullptr;
    double* dcur = buf;
    const double* din[14];
    for (int i = 0; i < 14; ++i) {
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(dcur, in_host[i], in_cnt[i] * sizeof(double), hipMemcpyHostToDevice, st));
        din[i] = dcur;
        dcur += in_cnt[i];
    }
    TRAJOPT_HIP_CHECK(hipMemsetAsync(dcur, 0, (out_total + 2) * sizeof(double), st));
    double* dout[6];
    for (int i = 0; i < 6; ++i) { dout[i] = dcur; dcur += out_cnt[i]; }
    double* d_lam = dcur;
    int32_t* d_div = reinterpret_cast<int32_t*>(dcur + 1);
    TRAJOPT_HIP_CHECK(hipMemcpyAsync(d_lam, &lmbda, sizeof(double), hipMemcpyHostToDevice, st));
    trajopt_bsp_backward_io_t io{};
    io.Q = din[0]; io.q = din[1]; io.R = din[2]; io.r = din[3]; io.P = din[4]; io.p = din[5]; io.F = din[6];
    io.G = din[7]; io.T = din[8]; io.U = din[9]; io.V = din[10]; io.X = din[11]; io.Y = din[12]; io.Z = din[13];
    io.lmbda = d_lam;
    io.S = dout[0]; io.s = dout[1]; io.tau = dout[2]; io.dS = dout[3]; io.K = dout[4]; io.kff = dout[5];
    io.diverge = d_div;
    const int rc = trajopt_bspilqr_backward_pass_batched(&io, reg, tau_mode, b, m, T, 1, TRAJOPT_LAYOUT_PROBLEM_MAJOR, st);
    if (rc != TRAJOPT_OK) return rc;
    for (int i = 0; i < 6; ++i)
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(out_host[i], dout[i], out_cnt[i] * sizeof(double), hipMemcpyDeviceToHost, st));
    int32_t div = -1;
    TRAJOPT_HIP_CHECK(hipMemcpyAsync(&div, d_div, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TRAJOPT_HIP_CHECK(hipStreamSynchronize(st));
    *diverge = div > 0 ? div : 0;   // util.cpp:139,171
    if (failed) *failed = div >= 0 ? 1 : 0;
    return TRAJOPT_OK;
}

}  // extern "C"
