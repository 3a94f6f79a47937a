// "System" policies: what the rollout / quadratisation kernels need from an env model, with either
// compile-time dimensions (the physical models) or run-time ones (the affine LQR env).
#pragma once
#include "env_models.h"

namespace trajopt {

template <int MODEL>
struct SysPhys {
    static constexpr int N = Model<MODEL>::N, M = Model<MODEL>::M;
    static constexpr bool kStatic = true;
    static __device__ __forceinline__ int n(const trajopt_env_t&) { return N; }
    static __device__ __forceinline__ int m(const trajopt_env_t&) { return M; }
    static __device__ __forceinline__ void dynamics(const trajopt_env_t& e, const double (&x)[N], const double (&u)[M],
                                                    double (&xn)[N]) {
        env_dynamics<MODEL, double>(e, x, u, xn);
    }
    // same for an action already clipped to +-umax
    static __device__ __forceinline__ void dynamics_clipped(const trajopt_env_t& e, const double (&x)[N],
                                                            const double (&u)[M], double (&xn)[N]) {
        env_dynamics<MODEL, double, false>(e, x, u, xn);
    }
    // ... and the state clip as two hardware instructions per state (env_models.h: clip_hw)
    static __device__ __forceinline__ void dynamics_clipped_hw(const trajopt_env_t& e, const double (&x)[N],
                                                               const double (&u)[M], double (&xn)[N]) {
        env_dynamics_hw<MODEL>(e, x, u, xn);
    }
    // Columns [k0, k0+KC) of the Jacobian [A | B] of env.dynamics at (x, u): J[i][k - k0].
    template <int KC>
    static __device__ __forceinline__ void jacobian_cols(const trajopt_env_t& e, const double (&x)[N],
                                                         const double (&u)[M], int k0, double (&J)[N][KC]) {
        constexpr int C = CyclicLeadingStates<MODEL>::value;
        if constexpr (C > 0 && C < KC) {
            // the chunk that holds the cyclic states' columns (k0 = 0; with all columns in one chunk the only one): KC - C
            // tangent directions, columns 0 .. C - 1 written down
            if (KC == N + M || k0 == 0) {
                constexpr int K1 = KC - C;
                Dual<K1> xd[N], ud[M], xn[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    xd[i] = constant<K1>(x[i]);
#pragma unroll
                    for (int k = 0; k < K1; ++k) xd[i].d[k] = (i == k + C) ? 1.0 : 0.0;
                }
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    ud[j] = constant<K1>(u[j]);
#pragma unroll
                    for (int k = 0; k < K1; ++k) ud[j].d[k] = (N + j == k + C) ? 1.0 : 0.0;
                }
                env_dynamics<MODEL, Dual<K1>>(e, xd, ud, xn);
#pragma unroll
                for (int i = 0; i < N; ++i) {
#pragma unroll
                    for (int c = 0; c < C; ++c)   // clip_g's mask
                        J[i][c] = (i == c && xn[c].v != -e.xmax[c] && xn[c].v != e.xmax[c]) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = 0; k < K1; ++k) J[i][k + C] = xn[i].d[k];
                }
                return;
            }
        }
        Dual<KC> xd[N], ud[M], xn[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            xd[i] = constant<KC>(x[i]);
#pragma unroll
            for (int k = 0; k < KC; ++k) xd[i].d[k] = (i == k0 + k) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            ud[j] = constant<KC>(u[j]);
#pragma unroll
            for (int k = 0; k < KC; ++k) ud[j].d[k] = (N + j == k0 + k) ? 1.0 : 0.0;
        }
        env_dynamics<MODEL, Dual<KC>>(e, xd, ud, xn);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < KC; ++k) J[i][k] = xn[i].d[k];
    }
};

struct SysLqr {
    static constexpr int N = TRAJOPT_ENV_MAX_STATE, M = TRAJOPT_ENV_MAX_ACT;
    static constexpr bool kStatic = false;
    static __device__ __forceinline__ int n(const trajopt_env_t& e) { return e.dm_state; }
    static __device__ __forceinline__ int m(const trajopt_env_t& e) { return e.dm_act; }
    static __device__ __forceinline__ void dynamics(const trajopt_env_t& e, const double (&x)[N], const double (&u)[M],
                                                    double (&xn)[N]) {
        lqr_dynamics<double>(e, x, u, xn);
    }
    static __device__ __forceinline__ void dynamics_clipped(const trajopt_env_t& e, const double (&x)[N],
                                                            const double (&u)[M], double (&xn)[N]) {
        lqr_dynamics<double>(e, x, u, xn);
    }
    template <int KC>
    static __device__ __forceinline__ void jacobian_cols(const trajopt_env_t& e, const double (&x)[N],
                                                         const double (&u)[M], int k0, double (&J)[N][KC]) {
        // d(A x + B clip(u) + c): A, and B masked where the clip is active (lqr_v0.py:50-53)
        const int nn = e.dm_state, mm = e.dm_act;
        for (int k = 0; k < KC; ++k) {
            const int col = k0 + k;
            for (int i = 0; i < nn; ++i) {
                double v = 0.0;
                if (col < nn) v = e.lin_A[i + nn * col];
                else if (col < nn + mm) {
                    const int j = col - nn;
                    const double c = clip_g(u[j], e.umax[j]);
                    v = (c != -e.umax[j] && c != e.umax[j]) ? e.lin_B[i + nn * j] : 0.0;
                }
                J[i][k] = v;
            }
        }
    }
};

// Run `fn.template operator()<SYS>()` for the env's model; returns false for an unknown model.
template <class F>
inline bool dispatch_model(const trajopt_env_t& e, F&& fn) {
    if (e.nb_features < 0 || e.nb_features > TRAJOPT_ENV_MAX_STATE) return false;
    for (int f = 0; f < e.nb_features; ++f)
        if ((e.features[f] & 0xff) >= e.dm_state || (e.features[f] >> 8) > TRAJOPT_FEAT_SIN || e.features[f] < 0) return false;
    switch (e.model) {
        case TRAJOPT_ENV_PENDULUM:
            if (e.dm_state != 2 || e.dm_act != 1) return false;
            fn.template operator()<SysPhys<TRAJOPT_ENV_PENDULUM>>(); return true;
        case TRAJOPT_ENV_CARTPOLE:
            if (e.dm_state != 4 || e.dm_act != 1) return false;
            fn.template operator()<SysPhys<TRAJOPT_ENV_CARTPOLE>>(); return true;
        case TRAJOPT_ENV_DOUBLE_CARTPOLE:
            if (e.dm_state != 6 || e.dm_act != 1) return false;
            fn.template operator()<SysPhys<TRAJOPT_ENV_DOUBLE_CARTPOLE>>(); return true;
        case TRAJOPT_ENV_DOUBLE_PENDULUM:
            if (e.dm_state != 4 || e.dm_act != 2) return false;
            fn.template operator()<SysPhys<TRAJOPT_ENV_DOUBLE_PENDULUM>>(); return true;
        case TRAJOPT_ENV_QUAD_PENDULUM:
            if (e.dm_state != 8 || e.dm_act != 4) return false;
            fn.template operator()<SysPhys<TRAJOPT_ENV_QUAD_PENDULUM>>(); return true;
        case TRAJOPT_ENV_ROBOT:
            if (e.dm_state != 4 || e.dm_act != 2) return false;
            fn.template operator()<SysPhys<TRAJOPT_ENV_ROBOT>>(); return true;
        case TRAJOPT_ENV_QUADROTOR:
            if (e.dm_state != 12 || e.dm_act != 4) return false;
            fn.template operator()<SysPhys<TRAJOPT_ENV_QUADROTOR>>(); return true;
        case TRAJOPT_ENV_LQR:
            if (e.dm_state < 1 || e.dm_state > TRAJOPT_ENV_MAX_STATE || e.dm_act < 1 || e.dm_act > TRAJOPT_ENV_MAX_ACT)
                return false;
            fn.template operator()<SysLqr>(); return true;
        default: return false;
    }
}

}  // namespace trajopt
