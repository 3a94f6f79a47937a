// Shared by the GPS kernels (gps_tpp.hip, gps_ops.hip): PROBLEM_MINOR element access and du <= 2 inverses.
#pragma once
#include "smallmat.h"

namespace trajopt {

// element e of slice t of problem p in a PROBLEM_MINOR array with E elements per slice
template <int R>
__device__ __forceinline__ void gld(double (&x)[R], const double* arr, size_t E, int t, size_t NP, size_t p) {
#pragma unroll
    for (int e = 0; e < R; ++e) x[e] = arr[((size_t)t * E + e) * NP + p];
}
template <int R>
__device__ __forceinline__ void gst(double* arr, size_t E, int t, size_t NP, size_t p, const double (&x)[R]) {
#pragma unroll
    for (int e = 0; e < R; ++e) arr[((size_t)t * E + e) * NP + p] = x[e];
}

// inverse and determinant of an M x M matrix (M <= 2), the general inv() of util.cpp:338; false when singular
template <int M>
__device__ __forceinline__ bool small_inv(const double (&X)[M * M], double (&Xi)[M * M], double& det) {
    static_assert(M == 1 || M == 2, "instantiated for du <= 2");
    if constexpr (M == 1) {
        det = X[0];
        Xi[0] = 1.0 / X[0];
    } else {
        det = X[0] * X[3] - X[2] * X[1];
        const double id = 1.0 / det;
        Xi[0] = X[3] * id; Xi[1] = -X[1] * id; Xi[2] = -X[2] * id; Xi[3] = X[0] * id;
    }
    return det != 0.0 && det == det;
}


// inv_sympd() of an M x M matrix (M <= 2) from its lower triangle, as Armadillo's Cholesky-based routine reads it;
// false when it is not positive definite (Armadillo throws).  det = its determinant.
template <int M>
__device__ __forceinline__ bool small_inv_sympd(const double (&X)[M * M], double (&Xi)[M * M], double& det) {
    static_assert(M == 1 || M == 2, "instantiated for du <= 2");
    if constexpr (M == 1) {
        det = X[0];
        Xi[0] = 1.0 / X[0];
        return X[0] > 0.0;
    } else {
        const double s2 = X[3] - X[1] * X[1] / X[0];   // second Cholesky pivot squared
        det = X[0] * X[3] - X[1] * X[1];
        const double id = 1.0 / det;
        Xi[0] = X[3] * id; Xi[1] = -X[1] * id; Xi[2] = -X[1] * id; Xi[3] = X[0] * id;
        return X[0] > 0.0 && s2 > 0.0;
    }
}

inline bool gps_shape_supported(int n, int m) {
    return (m == 1 && (n == 2 || n == 3 || n == 4 || n == 6)) || (m == 2 && (n == 2 || n == 4));
}

// gen_wave.hip: wave-per-problem kernels with run-time dimensions for the shapes without an instantiation
bool gps_generic_shape_ok(int n, int m);
int launch_gps_backward_generic(const trajopt_gps_backward_io_t& io, int n, int m, int T, int64_t NP, hipStream_t st);
int launch_gps_forward_generic(const double* mu_x0, const double* sigma_x0, const double* A, const double* B,
                               const double* c, const double* sigma_dyn, const double* K, const double* kff,
                               const double* sigma_ctl, int n, int m, int T, int64_t NP, double* mu_x, double* sigma_x,
                               double* mu_u, double* sigma_u, double* mu_xu, double* sigma_xu, hipStream_t st);
int launch_gps_augment_generic(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                               const double* c0, const double* K, const double* kff, const double* sigma_ctl,
                               const double* alpha, int n, int m, int T, int64_t NP, double* agCxx, double* agcx,
                               double* agCuu, double* agcu, double* agCxu, double* agc0, int32_t* status, hipStream_t st);
int launch_gps_kl_generic(const double* pK, const double* pk, const double* pS, const double* qK, const double* qk,
                          const double* qS, const double* mu_x, const double* sigma_x, int n, int m, int T, int64_t NP,
                          double* kl, int32_t* status, hipStream_t st);
int launch_gps_quadexp_generic(const double* mu, const double* sg, const double* Q, const double* q, const double* q0,
                               double* out, int n, int64_t NP, hipStream_t st);
bool bsp_generic_shape_ok(int b, int m);
int launch_bsp_backward_generic(const trajopt_bsp_backward_io_t& io, int reg, int tau_mode, int b, int m, int T,
                                int64_t NP, hipStream_t st);

// fn.template operator()<N, M>() for a supported shape
template <class F>
inline bool gps_dispatch(int n, int m, F&& fn) {
    if (m == 1) {
        if (n == 2) { fn.template operator()<2, 1>(); return true; }
        if (n == 3) { fn.template operator()<3, 1>(); return true; }
        if (n == 4) { fn.template operator()<4, 1>(); return true; }
        if (n == 6) { fn.template operator()<6, 1>(); return true; }
    } else if (m == 2) {
        if (n == 2) { fn.template operator()<2, 2>(); return true; }
        if (n == 4) { fn.template operator()<4, 2>(); return true; }
    }
    return false;
}

}  // namespace trajopt
