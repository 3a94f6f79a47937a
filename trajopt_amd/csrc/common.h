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
    // v_rcp_f64 is good to ~2^-23; one CUBIC step, r (1 + e + e^2) with e = 1 - x r, leaves e^3 ~ 2^-69 - the same
    // <= 1 ulp as two Newton steps in three dependent FMAs instead of four (these kernels are chains of dependent
    // instructions: the rollout takes four reciprocals a step, every backward pass at least one)
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- the batched driver's candidate planes (trajopt_ilqr_workspace_t::cand_state / cand_action, ABI v5) ------------
// Accepting a candidate copies nothing (ilqr.py:210-217): problem p's reference trajectory IS candidate plane
// ref_plane[p].  Neighbouring problems accept different step sizes, i.e. read different planes, so a plane must not
// put neighbouring problems into one cache line (a reader would use 8 bytes of every 128-byte line: measured, the
// rollout ran at half speed on the plain struct-of-arrays planes).  The planes are therefore TIME-BLOCKED: element e
// of steps 8 b .. 8 b + 7 of one problem is one 64-byte run,
//     plane[(((t / 8) * rows + e) * nb_problems + p) * 8 + t % 8],      plane size = ceil(slices / 8) * 8 * rows * nb_problems,
// which a reader of that problem uses in full over 8 consecutive steps, whatever plane its neighbours read.
constexpr int kTB = 8;
__host__ __device__ inline size_t tb_blocks(int slices) { return (size_t)(slices + kTB - 1) / kTB; }
__host__ __device__ inline size_t tb_plane(int rows, int slices, size_t NP) { return tb_blocks(slices) * kTB * (size_t)rows * NP; }
__host__ __device__ inline size_t tb_index(int rows, int t, int e, size_t p, size_t NP) {
    return ((((size_t)(t / kTB) * rows + e) * NP + p) * kTB) + (size_t)(t % kTB);
}

// Where the driver's quadratisation finds the reference trajectory: states / actions of problem p in time-blocked
// plane ref_plane[p] of cand_state / cand_action, or - ref_plane[p] < 0 - in the dense xref / uref the kernel is given.
// What it reads from a plane is written through to xref_out / uref_out (the workspace's dense PROBLEM_MINOR xref /
// uref), from which the backward pass (gradient norm, ilqr.py:176) and the rollout - whose lanes of one problem sit in
// eleven different waves - take the reference with wave-uniform, coalesced addressing, and where it survives when the
// plane is written again.
struct RefView {
    const int32_t* ref_plane;  // NULL: the dense xref / uref for every problem
    const double *cand_state, *cand_action;
    double *xref_out, *uref_out;
};

// The rollout's side: its candidates go, time-blocked, into planes [first_plane, first_plane + nb_alphas) of
// cand_state / cand_action - the bank of this iteration - and the lanes of stopped problems (which run along, see
// rollout.hip) into junk_plane; the gains of problem p come from bank out_bank[p] (trajopt_backward_io_t::out_bank).
struct RolloutPlanes {
    int first_plane, junk_plane;
    const int32_t* out_bank;   // bank of the multi-buffered gains each problem reads (NULL: plain K / kff)
    // Two-pass line search (driver.hip): pass 1 = cost_only (returns of every alpha, no trajectory written); pass 2 =
    // sel given: one lane per problem rolls alpha sel[p] (>= 0) out again and writes it to plane first_plane
    int cost_only;
    const int32_t* sel;
};

// The lambda schedule between iterations (ilqr.py:166-167 / :178-179 and :207-208 / :226-227).
__device__ __forceinline__ void grow_lambda(double& lam, double& dlam, const trajopt_lambda_schedule_t& s) {
    dlam = fmax(dlam * s.mult_lmbda, s.mult_lmbda);  // ilqr.py:166 / :226
    lam = fmax(lam * dlam, s.min_lmbda);             // ilqr.py:167 / :227
}
__device__ __forceinline__ void shrink_lambda(double& lam, double& dlam, const trajopt_lambda_schedule_t& s) {
    dlam = fmin(dlam / s.mult_lmbda, 1.0 / s.mult_lmbda);  // ilqr.py:178 / :207
    lam = lam * dlam * (lam > s.min_lmbda ? 1.0 : 0.0);    // ilqr.py:179 / :208
}

// What the driver does per problem right after the backward pass (ilqr.py:175-180, :168 -> :224-231; post_backward_kernel
// in driver.hip), for a backward kernel that does it itself in the lane that writes the problem's scalars: one launch
// (~9 us of a 0.68 ms iteration at C2) less.  on = 0: the driver launches post_backward_kernel.
struct PostBackward {
    int32_t *ref_plane, *status, *active;
    double tolgrad;
    int nb_alphas, wb, on;
};
__device__ __forceinline__ void post_backward_problem(const PostBackward& pb, const trajopt_lambda_schedule_t& sched, size_t p,
                                                      double& lam, double& dlam, double g_norm, int diverge) {
    const int ref = pb.ref_plane[p];
    if (ref >= 0 && ref / pb.nb_alphas == pb.wb) pb.ref_plane[p] = -1;
    if (g_norm < pb.tolgrad && lam < 1e-5) {
        shrink_lambda(lam, dlam, sched);
        pb.status[p] = 1;
        pb.active[p] = 0;
    } else if (diverge >= 0) {
        grow_lambda(lam, dlam, sched);
        if (lam > sched.max_lmbda) pb.status[p] = 3;
        pb.active[p] = 0;
    }
}

// Bank of the multi-buffered gains problem p writes (trajopt_backward_io_t::out_bank), 0 without banks.
__device__ __forceinline__ int gains_bank(const int32_t* out_bank, size_t p) { return out_bank ? out_bank[p] : 0; }

// Kernel launchers implemented per translation unit (all return TRAJOPT_* codes).
int launch_backward_tpp(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int retry,
                        const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled);
bool col8_shape_ok(int n, int m, int T, int64_t NP);
int launch_backward_col8(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int retry,
                         const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled);
int launch_backward_mfma4(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int retry,
                          const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled);
int launch_backward_mfma16(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int layout,
                           int retry, const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled);
int launch_backward_wg(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int layout,
                       int retry, const trajopt_lambda_schedule_t& sched, hipStream_t st);
int launch_riccati_wg(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                      const double* A, const double* B, const double* c, int n, int m, int T, int64_t NP,
                      int layout, double* V, double* v, double* K, double* kff, int32_t* status, hipStream_t st,
                      const int32_t* only = nullptr);   // only: run the problems with only[p] != 0

int launch_riccati_mfma64(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                          const double* A, const double* B, const double* c, int n, int m, int T, int64_t NP,
                          int layout, double* V, double* v, double* K, double* kff, int32_t* status, hipStream_t st,
                          bool* handled);

}  // namespace trajopt
