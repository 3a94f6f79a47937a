// Quadratisation along a reference trajectory (rows A4, A5, A10).
//
//   AnalyticalLinearDynamics.taylor_expansion  /root/reference/trajopt/ilqr/objects.py:114-117
//   AnalyticalQuadraticCost.taylor_expansion   /root/reference/trajopt/ilqr/objects.py:68-78
//   Riccati convention                         /root/reference/trajopt/riccati/objects.py:54-65,102-108
//
// Embarrassingly parallel over (problem, time step): one thread per (problem, step, tangent chunk).
// A chunk is KC columns of the Jacobian [A | B]; the thread pushes KC tangent directions through
// clip -> RK4 -> clip with dual numbers (exact derivatives, autograd's clip rule).  KC trades register
// pressure (a Dual<KC> is KC+1 doubles) against recomputing the primal once per chunk.
// The cost expansion is closed-form because every env in scope has identity features:
//   Cxx = 2 a diag(gw), cx = 2 a gw (y - g), Cuu = 2 diag(uw), cu = 2 uw (u - [u_last if slew]), Cxu = 0,
// all times dt for the LQR envs, and the state terms vanish when the weight a is exactly 0.
#include "sys.h"

namespace trajopt {


struct QuadArgs {
    trajopt_env_t env;
    const double *xref, *uref, *weighting;
    const int32_t* active;
    double *A, *B, *c, *Cxx, *cx, *Cuu, *cu, *Cxu;
    size_t NP;
    int T, convention;
    // Driver only (no copy on accept, RefView in common.h): with `ref_plane` non-NULL problem p reads its reference
    // trajectory from the time-blocked candidate plane ref_plane[p] (>= 0; else from the dense xref / uref) and writes
    // what it read from a plane through to xref_out / uref_out - the dense arrays the backward pass (gradient norm,
    // ilqr.py:176) and the rollout take it from.
    const int32_t* ref_plane;
    const double *cand_state, *cand_action;
    double *xref_out, *uref_out;
};

template <bool MINOR>
__device__ __forceinline__ size_t qi(size_t e_total, size_t E_total, size_t p, size_t NP) {
    return MINOR ? e_total * NP + p : p * E_total + e_total;
}

// MINOR: layout of xref / uref (and of the candidate planes);  OMINOR: layout of A, B, C.., and of K / kff.
template <class SYS, int KC, bool MINOR, bool OMINOR>
__global__ __launch_bounds__(64) void quadratize_kernel(QuadArgs a) {
    constexpr int N = SYS::N, M = SYS::M;
    const trajopt_env_t& e = a.env;
    const int n = SYS::n(e), m = SYS::m(e), T = a.T;
    const size_t NP = a.NP;
    // grid = (problem blocks, T + 1, tangent chunks): no 64-bit division in the index math
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = (int)blockIdx.y, chunk = (int)blockIdx.z;
    // Lanes without a problem (p >= NP, or one the driver has stopped) compute nothing - but where a wave's chunk goes out
    // through the LDS stage (kStage, below) they still help to write their neighbours' columns.
    const bool live = p < NP && !(a.active && a.active[p < NP ? p : 0] == 0);
    const size_t En1 = (size_t)n * (T + 1), Em = (size_t)m * T;
    typedef double q_v2 __attribute__((ext_vector_type(2)));
    // kStage (PROBLEM_MAJOR outputs, static even dx, chunks that do not straddle A | B: quadrotor, quad pendulum): a lane's
    // chunk is ONE contiguous run of KC * dx doubles of its (problem, step) - written straight from the lanes every 16-byte
    // store of the wave lands in 64 different cache lines, and the kernel is bound by exactly that.
    // So the wave's 64 runs go through LDS and out again with consecutive lanes on consecutive 16-byte pieces: whole lines.
    constexpr bool kStage = !OMINOR && SYS::kStatic && N % 2 == 0 && N % KC == 0 && (N + M) % KC == 0;
    constexpr int kPieces = KC * N / 2;   // 16-byte pieces of a lane's run
    __shared__ q_v2 s_stage[kStage ? 64 * kPieces : 1];
    bool staged = false;
    if (live) {
        double x[N], u[M], ul[M];
        bool through = false;
        const int k = a.ref_plane ? a.ref_plane[p] : -1;
        if (k >= 0) {   // the driver: time-blocked candidate plane ref_plane[p] (common.h)
            const double* xs = a.cand_state + (size_t)k * tb_plane(n, T + 1, NP);
            const double* us = a.cand_action + (size_t)k * tb_plane(m, T, NP);
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = (i < n) ? xs[tb_index(n, t, i, p, NP)] : 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                u[j] = (j < m && t < T) ? us[tb_index(m, t, j, p, NP)] : 0.0;
                ul[j] = (j < m && t >= 1) ? us[tb_index(m, t - 1, j, p, NP)] : 0.0;
            }
            through = (chunk == 0 && a.uref_out != nullptr && a.xref_out != nullptr);
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = (i < n) ? a.xref[qi<MINOR>((size_t)n * t + i, En1, p, NP)] : 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                // _u = hstack(u, 0): the pad is u_T and, through Python's negative index, also u_last at t = 0
                u[j] = (j < m && t < T) ? a.uref[qi<MINOR>((size_t)m * t + j, Em, p, NP)] : 0.0;
                ul[j] = (j < m && t >= 1) ? a.uref[qi<MINOR>((size_t)m * (t - 1) + j, Em, p, NP)] : 0.0;
            }
        }
        if (through) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i < n) a.xref_out[qi<MINOR>((size_t)n * t + i, En1, p, NP)] = x[i];
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j < m && t < T) a.uref_out[qi<MINOR>((size_t)m * t + j, Em, p, NP)] = u[j];
        }

        if (t < T) {  // dynamics Jacobian columns of this chunk
            const int k0 = chunk * KC;
            double J[N][KC];
            SYS::template jacobian_cols<KC>(e, x, u, k0, J);
            const size_t Enn = (size_t)n * n * T, Enm = (size_t)n * m * T;
            // PROBLEM_MAJOR outputs with an even, static dx (quadrotor): a column of a step is dx consecutive doubles at a
            // 16-byte aligned address - written two rows a store (half as many store instructions, each of which touches
            // 64 different cache lines whatever its width)
            constexpr bool kPair = !OMINOR && SYS::kStatic && N % 2 == 0;
            if constexpr (kStage) {
#pragma unroll
                for (int k = 0; k < KC; ++k)
#pragma unroll
                    for (int i = 0; i < N; i += 2)
                        s_stage[(int)threadIdx.x * kPieces + k * (N / 2) + i / 2] = q_v2{J[i][k], J[i + 1][k]};
                staged = true;
            } else
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int col = k0 + k;
                if (col < n) {
                    double* dst = a.A + qi<OMINOR>((size_t)n * n * t + (size_t)n * col, Enn, p, NP);
                    if constexpr (kPair) {
#pragma unroll
                        for (int i = 0; i < N; i += 2) *reinterpret_cast<q_v2*>(dst + i) = q_v2{J[i][k], J[i + 1][k]};
                    } else {
#pragma unroll
                        for (int i = 0; i < N; ++i)
                            if (i < n) a.A[qi<OMINOR>((size_t)n * n * t + i + (size_t)n * col, Enn, p, NP)] = J[i][k];
                    }
                } else if (col < n + m) {
                    double* dst = a.B + qi<OMINOR>((size_t)n * m * t + (size_t)n * (col - n), Enm, p, NP);
                    if constexpr (kPair) {
#pragma unroll
                        for (int i = 0; i < N; i += 2) *reinterpret_cast<q_v2*>(dst + i) = q_v2{J[i][k], J[i + 1][k]};
                    } else {
#pragma unroll
                        for (int i = 0; i < N; ++i)
                            if (i < n) a.B[qi<OMINOR>((size_t)n * m * t + i + (size_t)n * (col - n), Enm, p, NP)] = J[i][k];
                    }
                }
            }
        }
        if (chunk == 0) {  // cost expansion (objects.py:72-78)
            const double w = a.weighting[t];
            const double sc = e.cost_times_dt ? e.dt : 1.0;
            const size_t Enn1 = (size_t)n * n * (T + 1), Emm1 = (size_t)m * m * (T + 1), Enm1 = (size_t)n * m * (T + 1),
                         Em1 = (size_t)m * (T + 1);
            double fgx[N], fhxx[N];
            if (e.nb_features != 0) feature_cost<N, true>(e, x, n, fgx, fhxx);
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i < n) {
                    double g = 0.0, h = 0.0;
                    if (w != 0.0) {
                        if (e.nb_features == 0) {
                            const double y = ((e.periodic >> i) & 1) ? wrap_angle(x[i]) : x[i];
                            h = w * (sc * 2.0 * e.gw[i]);  // = w[t] * cost_gxx[i], the structured form the backward pass can take
                            g = h * (y - e.goal[i]);
                        } else {   // frozen feature Jacobian (cartpole.py:122-124): Gauss-Newton
                            h = w * (sc * 2.0) * fhxx[i];
                            g = w * (sc * 2.0) * fgx[i];
                        }
                    }
                    a.cx[qi<OMINOR>((size_t)n * t + i, En1, p, NP)] = g;
                    if (a.Cxx) {  // NULL: the backward pass forms Cxx, Cuu, Cxu itself (trajopt_backward_io_t::cost_w)
#pragma unroll
                        for (int j = 0; j < N; ++j)
                            if (j < n) a.Cxx[qi<OMINOR>((size_t)n * n * t + j + (size_t)n * i, Enn1, p, NP)] = (i == j) ? h : 0.0;
#pragma unroll
                        for (int j = 0; j < M; ++j)
                            if (j < m) a.Cxu[qi<OMINOR>((size_t)n * m * t + i + (size_t)n * j, Enm1, p, NP)] = 0.0;
                    }
                }
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j < m) {
                    const double d = e.slew_rate ? (u[j] - ul[j]) : u[j];
                    a.cu[qi<OMINOR>((size_t)m * t + j, Em1, p, NP)] = sc * 2.0 * e.uw[j] * d;
                    if (a.Cxx) {
#pragma unroll
                        for (int k = 0; k < M; ++k)
                            if (k < m)
                                a.Cuu[qi<OMINOR>((size_t)m * m * t + k + (size_t)m * j, Emm1, p, NP)] =
                                    (k == j) ? sc * 2.0 * e.uw[j] : 0.0;
                    }
                }
        }
    }
    if constexpr (kStage) {
        if (t < T) {   // (uniform: one step a block)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned long long okm = __ballot(staged);
            const int k0 = chunk * KC, lane = (int)threadIdx.x;
            const size_t p0 = (size_t)blockIdx.x * 64;
            const bool isA = k0 < n;
            const size_t E = isA ? (size_t)n * n * T : (size_t)n * m * T;
            double* base = isA ? a.A + (size_t)n * n * t + (size_t)n * k0 : a.B + (size_t)n * m * t + (size_t)n * (k0 - n);
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int g = i * 64 + lane, src = g / kPieces, piece = g % kPieces;   // the lane that computed it, piece of its run
                if ((okm >> src) & 1ull) *reinterpret_cast<q_v2*>(base + (p0 + (size_t)src) * E + 2 * piece) = s_stage[g];
            }
        }
    }
}

// Riccati convention fix-up (riccati/objects.py:60-65,107-108), one thread per (problem, step):
//   c_t = f(x,u) - A x - B u;  cx -= Cxx x + 2 Cxu u;  cu -= Cuu u + 2 Cxu^T x;  Cxx, Cuu, Cxu *= 1/2.
template <class SYS, bool MINOR>
__global__ __launch_bounds__(64) void riccati_fixup_kernel(QuadArgs a) {
    constexpr int N = SYS::N, M = SYS::M;
    const trajopt_env_t& e = a.env;
    const int n = SYS::n(e), m = SYS::m(e), T = a.T;
    const size_t NP = a.NP;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = (int)blockIdx.y;
    if (p >= NP) return;
    if (a.active && a.active[p] == 0) return;
    const size_t En1 = (size_t)n * (T + 1), Em = (size_t)m * T, En = (size_t)n * T;
    const size_t Enn = (size_t)n * n * T, Enm = (size_t)n * m * T;
    const size_t Enn1 = (size_t)n * n * (T + 1), Emm1 = (size_t)m * m * (T + 1), Enm1 = (size_t)n * m * (T + 1),
                 Em1 = (size_t)m * (T + 1);
    double x[N], u[M];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = (i < n) ? a.xref[qi<MINOR>((size_t)n * t + i, En1, p, NP)] : 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) u[j] = (j < m && t < T) ? a.uref[qi<MINOR>((size_t)m * t + j, Em, p, NP)] : 0.0;
    if (t < T) {
        double f[N];
        SYS::dynamics(e, x, u, f);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < n) {
                double ax = 0.0, bu = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j < n) ax = fma(a.A[qi<MINOR>((size_t)n * n * t + i + (size_t)n * j, Enn, p, NP)], x[j], ax);
#pragma unroll
                for (int j = 0; j < M; ++j)
                    if (j < m) bu = fma(a.B[qi<MINOR>((size_t)n * m * t + i + (size_t)n * j, Enm, p, NP)], u[j], bu);
                a.c[qi<MINOR>((size_t)n * t + i, En, p, NP)] = f[i] - ax - bu;
            }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < n) {
            double hx = 0.0, pu = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (j < n) hx = fma(a.Cxx[qi<MINOR>((size_t)n * n * t + i + (size_t)n * j, Enn1, p, NP)], x[j], hx);
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j < m) pu = fma(a.Cxu[qi<MINOR>((size_t)n * m * t + i + (size_t)n * j, Enm1, p, NP)], u[j], pu);
            const size_t o = qi<MINOR>((size_t)n * t + i, En1, p, NP);
            a.cx[o] = a.cx[o] - hx - 2.0 * pu;
        }
#pragma unroll
    for (int j = 0; j < M; ++j)
        if (j < m) {
            double ru = 0.0, xp = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k)
                if (k < m) ru = fma(a.Cuu[qi<MINOR>((size_t)m * m * t + j + (size_t)m * k, Emm1, p, NP)], u[k], ru);
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i < n) xp = fma(x[i], a.Cxu[qi<MINOR>((size_t)n * m * t + i + (size_t)n * j, Enm1, p, NP)], xp);
            const size_t o = qi<MINOR>((size_t)m * t + j, Em1, p, NP);
            a.cu[o] = a.cu[o] - ru - 2.0 * xp;
        }
    for (int k = 0; k < n * n; ++k) a.Cxx[qi<MINOR>((size_t)n * n * t + k, Enn1, p, NP)] *= 0.5;
    for (int k = 0; k < m * m; ++k) a.Cuu[qi<MINOR>((size_t)m * m * t + k, Emm1, p, NP)] *= 0.5;
    for (int k = 0; k < n * m; ++k) a.Cxu[qi<MINOR>((size_t)n * m * t + k, Enm1, p, NP)] *= 0.5;
}

template <class SYS> struct ChunkOf { static constexpr int KC = 4; };
template <> struct ChunkOf<SysPhys<TRAJOPT_ENV_PENDULUM>> { static constexpr int KC = 3; };
template <> struct ChunkOf<SysPhys<TRAJOPT_ENV_CARTPOLE>> { static constexpr int KC = 5; };
template <> struct ChunkOf<SysPhys<TRAJOPT_ENV_DOUBLE_CARTPOLE>> { static constexpr int KC = 7; };   // all columns in one pass: 0.88 ms at C3 (KC = 4: 1.07, KC = 2: 1.48)
template <> struct ChunkOf<SysPhys<TRAJOPT_ENV_QUADROTOR>> { static constexpr int KC = 4; };
template <> struct ChunkOf<SysPhys<TRAJOPT_ENV_DOUBLE_PENDULUM>> { static constexpr int KC = 3; };
template <> struct ChunkOf<SysPhys<TRAJOPT_ENV_ROBOT>> { static constexpr int KC = 3; };
template <> struct ChunkOf<SysPhys<TRAJOPT_ENV_QUAD_PENDULUM>> { static constexpr int KC = 4; };

// the Riccati fix-up kernel reads A, B, C.. with the input layout flag: mixed layouts are an iLQR-driver feature
static inline bool a_convention_needs_same_layout(int convention, int layout, int out_layout) {
    return convention == 1 && layout != out_layout;
}

struct QuadLaunch {
    QuadArgs a;
    int layout, out_layout;
    hipStream_t st;
    template <class SYS>
    void operator()() {
        constexpr int KC = ChunkOf<SYS>::KC;
        const int n = a.env.dm_state, m = a.env.dm_act;
        const int nchunk = (n + m + KC - 1) / KC;
        const dim3 grid((unsigned)((a.NP + 63) / 64), (unsigned)(a.T + 1), (unsigned)nchunk), block(64);
        const bool minor = layout == TRAJOPT_LAYOUT_PROBLEM_MINOR, ominor = out_layout == TRAJOPT_LAYOUT_PROBLEM_MINOR;
        if (minor && ominor) hipLaunchKernelGGL((quadratize_kernel<SYS, KC, true, true>), grid, block, 0, st, a);
        else if (minor) hipLaunchKernelGGL((quadratize_kernel<SYS, KC, true, false>), grid, block, 0, st, a);
        else if (ominor) hipLaunchKernelGGL((quadratize_kernel<SYS, KC, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((quadratize_kernel<SYS, KC, false, false>), grid, block, 0, st, a);
        if (a.convention == 1) {
            const dim3 grid2((unsigned)((a.NP + 63) / 64), (unsigned)(a.T + 1));
            if (minor) hipLaunchKernelGGL((riccati_fixup_kernel<SYS, true>), grid2, block, 0, st, a);
            else hipLaunchKernelGGL((riccati_fixup_kernel<SYS, false>), grid2, block, 0, st, a);
        }
    }
};

int launch_quadratize(const trajopt_env_t* env, int T, int64_t NP, int layout, int convention, const double* xref,
                      const double* uref, const double* weighting, const int32_t* active, double* A, double* B,
                      double* c, double* Cxx, double* cx, double* Cuu, double* cu, double* Cxu, hipStream_t st,
                      const RefView* rv, int out_layout) {
    QuadLaunch L;
    L.a.env = *env;
    L.a.ref_plane = nullptr; L.a.cand_state = L.a.cand_action = nullptr; L.a.xref_out = L.a.uref_out = nullptr;
    if (rv && rv->ref_plane) {
        if (layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
        L.a.ref_plane = rv->ref_plane; L.a.cand_state = rv->cand_state; L.a.cand_action = rv->cand_action;
        L.a.xref_out = rv->xref_out; L.a.uref_out = rv->uref_out;
    }
    L.a.xref = xref; L.a.uref = uref; L.a.weighting = weighting; L.a.active = active;
    L.a.A = A; L.a.B = B; L.a.c = c; L.a.Cxx = Cxx; L.a.cx = cx; L.a.Cuu = Cuu; L.a.cu = cu; L.a.Cxu = Cxu;
    L.a.NP = (size_t)NP; L.a.T = T; L.a.convention = convention;
    L.layout = layout; L.out_layout = out_layout; L.st = st;
    if (a_convention_needs_same_layout(convention, layout, out_layout)) return TRAJOPT_ERR_BAD_ARG;
    if (!dispatch_model(*env, L)) return TRAJOPT_ERR_UNSUPPORTED;
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

}  // namespace trajopt

using namespace trajopt;

extern "C" int trajopt_quadratize_batched(const trajopt_env_t* env, int nb_steps, int64_t nb_problems, int layout,
                                          int convention, const double* xref, const double* uref,
                                          const double* weighting, const int32_t* active, double* A, double* B,
                                          double* c, double* Cxx, double* cx, double* Cuu, double* cu, double* Cxu,
                                          void* stream) {
    if (!env || !xref || !uref || !weighting || !A || !B || !Cxx || !cx || !Cuu || !cu || !Cxu) return TRAJOPT_ERR_BAD_ARG;
    if (nb_steps <= 0 || nb_problems <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
    if (convention != 0 && convention != 1) return TRAJOPT_ERR_BAD_ARG;
    if (convention == 1 && !c) return TRAJOPT_ERR_BAD_ARG;
    return launch_quadratize(env, nb_steps, nb_problems, layout, convention, xref, uref, weighting, active, A, B, c, Cxx,
                             cx, Cuu, cu, Cxu, as_stream(stream), nullptr, layout);
}
