// Register-resident tiny dense algebra for the thread-per-problem kernels.  Every loop bound is a
// template constant so hipcc unrolls fully and the arrays live in VGPRs (no runtime indexing).
#pragma once
#include "common.h"

namespace trajopt {

// Positive-definiteness test of the M x M matrix X (column-major) with the tolerance rule Armadillo's
// is_sympd() applies at /root/reference/trajopt/ilqr/src/util.cpp:145:
//   tol = 100 eps ||X||_F;  ||X - X^T||_inf / ||X||_inf <= tol;  chol(X - tol I) succeeds.
// On success L (lower, column-major) holds the Cholesky factor of X itself (not of X - tol I), which
// then solves for the gains (util.cpp:150-152 uses inv()).
template <int M>
__device__ __forceinline__ bool sympd_factor(const double (&X)[M * M], double (&L)[M * M]) {
    if constexpr (M == 1) {
        const double q = X[0];
        const double tol = 100.0 * kEps * fabs(q);
        if (!(q - tol > 0.0)) return false;
        L[0] = sqrt(q);
        return true;
    } else {
        double fro = 0.0;
#pragma unroll
        for (int k = 0; k < M * M; ++k) fro = fma(X[k], X[k], fro);
        fro = sqrt(fro);
        const double tol = 100.0 * kEps * fro;
        double na = 0.0, nd = 0.0;
        bool nan_seen = false;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double ra = 0.0, rd = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                ra += fabs(X[i + M * j]);
                rd += fabs(X[i + M * j] - X[j + M * i]);
            }
            nan_seen = nan_seen || (ra != ra);
            na = fmax(na, ra);
            nd = fmax(nd, rd);
        }
        if (nan_seen) return false;
        if (na != 0.0 && !((nd / na) <= tol)) return false;
        // pivots of X - tol I must all be positive
        {
            double Y[M * M];
#pragma unroll
            for (int k = 0; k < M * M; ++k) Y[k] = X[k];
#pragma unroll
            for (int i = 0; i < M; ++i) Y[i + M * i] -= tol;
            bool ok = true;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double d = Y[j + M * j];
#pragma unroll
                for (int k = 0; k < j; ++k) d = fma(-Y[j + M * k], Y[j + M * k], d);
                ok = ok && (d > 0.0);
                d = sqrt(d);
                Y[j + M * j] = d;
                const double inv = 1.0 / d;
#pragma unroll
                for (int i = j + 1; i < M; ++i) {
                    double s = Y[i + M * j];
#pragma unroll
                    for (int k = 0; k < j; ++k) s = fma(-Y[i + M * k], Y[j + M * k], s);
                    Y[i + M * j] = s * inv;
                }
            }
            if (!ok) return false;
        }
#pragma unroll
        for (int k = 0; k < M * M; ++k) L[k] = X[k];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double d = L[j + M * j];
#pragma unroll
            for (int k = 0; k < j; ++k) d = fma(-L[j + M * k], L[j + M * k], d);
            d = sqrt(d);
            L[j + M * j] = d;
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                double s = L[i + M * j];
#pragma unroll
                for (int k = 0; k < j; ++k) s = fma(-L[i + M * k], L[j + M * k], s);
                L[i + M * j] = s / d;
            }
        }
        return true;
    }
}

// x := (L L^T)^{-1} x for one right-hand side of length M.
template <int M>
__device__ __forceinline__ void chol_solve(const double (&L)[M * M], double (&x)[M]) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = fma(-L[i + M * k], x[k], s);
        x[i] = s / L[i + M * i];
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < M; ++k) s = fma(-L[k + M * i], x[k], s);
        x[i] = s / L[i + M * i];
    }
}

// Same PD rule as sympd_factor() (Armadillo's is_sympd, util.cpp:145) but square-root-free and with every
// division turned into one refined reciprocal per pivot: X = L D L' with unit lower L and id[j] = 1 / D[j].
// The test factorises X - tol I (pivots must be > 0), the returned factors are those of X itself.
template <int M>
__device__ __forceinline__ bool sympd_ldl(const double (&X)[M * M], double (&L)[M * M], double (&id)[M]) {
    double tol;
    if constexpr (M == 1) {
        tol = 100.0 * kEps * fabs(X[0]);
        if (!(X[0] - tol > 0.0)) return false;
        id[0] = rcp_nr(X[0]);
        L[0] = 1.0;
        return true;
    } else {
        double fro = 0.0;
#pragma unroll
        for (int k = 0; k < M * M; ++k) fro = fma(X[k], X[k], fro);
        tol = 100.0 * kEps * sqrt(fro);
        double na = 0.0, nd = 0.0;
        bool nan_seen = false;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double ra = 0.0, rd = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                ra += fabs(X[i + M * j]);
                rd += fabs(X[i + M * j] - X[j + M * i]);
            }
            nan_seen = nan_seen || (ra != ra);
            na = fmax(na, ra);
            nd = fmax(nd, rd);
        }
        if (nan_seen) return false;
        if (na != 0.0 && !(nd <= tol * na)) return false;
        bool ok = true;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {  // pass 0: X - tol I (the test), pass 1: X (the factors kept)
            double W[M * M];                    // W = L D (unscaled columns)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double dj = X[j + M * j] - (pass == 0 ? tol : 0.0);
#pragma unroll
                for (int k = 0; k < j; ++k) dj = fma(-W[j + M * k], L[j + M * k], dj);
                ok = ok && (dj > 0.0);
                id[j] = rcp_nr(dj);
                L[j + M * j] = 1.0;
#pragma unroll
                for (int i = j + 1; i < M; ++i) {
                    double w = X[i + M * j];
#pragma unroll
                    for (int k = 0; k < j; ++k) w = fma(-W[i + M * k], L[j + M * k], w);
                    W[i + M * j] = w;
                    L[i + M * j] = w * id[j];
                }
            }
            if (!ok) return false;
        }
        return true;
    }
}

// x := X^{-1} x with the factors of sympd_ldl (multiplications only).
template <int M>
__device__ __forceinline__ void ldl_solve(const double (&L)[M * M], const double (&id)[M], double (&x)[M]) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) x[i] = fma(-L[i + M * k], x[k], x[i]);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] *= id[i];
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
#pragma unroll
        for (int k = i + 1; k < M; ++k) x[i] = fma(-L[k + M * i], x[k], x[i]);
    }
}

// np.max / Armadillo-free NaN-propagating max used for the gradient norm (ilqr.py:176).
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? (a + b) : fmax(a, b); }

}  // namespace trajopt
