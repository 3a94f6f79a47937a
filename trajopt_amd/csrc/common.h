// Shared host/device helpers of the HIP library.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/trajopt_amd.h"

namespace trajopt {

// last HIP error text, for trajopt_last_hip_error()
void set_hip_error(hipError_t e, const char* where);

#define TRAJOPT_HIP_CHECK(expr)                                  \
    do {                                                         \
        hipError_t e__ = (expr);                                 \
        if (e__ != hipSuccess) {                                 \
            ::trajopt::set_hip_error(e__, #expr);                \
            return TRAJOPT_ERR_HIP;                              \
        }                                                        \
    } while (0)

// Offsets of element e (column-major, time slowest: e = i + rows*j + rows*cols*t) of problem p.
// PROBLEM_MINOR: e*NP + p.  PROBLEM_MAJOR: p*E + e  (E = elements per problem).
template <bool MINOR>
__host__ __device__ inline size_t elem_off(size_t e, size_t p, size_t NP, size_t E) {
    return MINOR ? e * NP + p : p * E + e;
}

constexpr double kEps = 2.220446049250313e-16;
constexpr int kMaxLambdaRetries = 128;  // lambda grows super-exponentially: 1e-6 -> 1e300 takes < 70 retries

// 1 / x to ~1 ulp from the hardware estimate and two Newton steps: ~45 cycles instead of the ~80 of an IEEE
// division (no scaling / fix-up: x must be a normal, finite, non-zero number - here always a positive pivot).
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Fused commit of the previous iteration's line search (driver only; ilqr.py:210-217): where sel[p] >= 0 the reference
// trajectory is read from candidate plane sel[p] and written through to xref / uref, and the candidate gains are promoted
// to K / kff.  Used by the quadratisation kernel and by the fused quadratise + backward kernel.
struct QuadCommit {
    const int32_t* sel;
    const double *cand_state, *cand_action, *cand_K, *cand_kff;
    double *xref, *uref, *K, *kff;
};

// Kernel launchers implemented per translation unit (all return TRAJOPT_* codes).
int launch_backward_tpp(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int retry,
                        const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled);
int launch_backward_mfma4(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int retry,
                          const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled);
int launch_backward_mfma16(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int layout,
                           int retry, const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled);
int launch_backward_wg(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int layout,
                       int retry, const trajopt_lambda_schedule_t& sched, hipStream_t st);
int launch_riccati_wg(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                      const double* A, const double* B, const double* c, int n, int m, int T, int64_t NP,
                      int layout, double* V, double* v, double* K, double* kff, int32_t* status, hipStream_t st);

int launch_riccati_mfma64(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                          const double* A, const double* B, const double* c, int n, int m, int T, int64_t NP,
                          int layout, double* V, double* v, double* K, double* kff, int32_t* status, hipStream_t st,
                          bool* handled);

}  // namespace trajopt
