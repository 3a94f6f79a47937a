/*
 * trajopt_amd.h - C ABI of the MI355X-native batched iLQR / Riccati engine.
 *
 * This is the drop-in boundary for the hot path of hanyas/trajopt: the pybind11/Armadillo module
 * `trajopt.ilqr.core` (one function, /root/reference/trajopt/ilqr/src/util.cpp:83-188) plus the
 * Python loops around it (/root/reference/trajopt/ilqr/ilqr.py:89-237,
 * /root/reference/trajopt/riccati/riccati.py:100-126).  Plain pointers and sizes only.
 *
 * Conventions
 *   - All real data is IEEE double.  Matrices are column-major; time is the slowest per-problem axis,
 *     exactly the reference's f_style arrays:  X(i, j, t) = X[i + rows*j + rows*cols*t]  (util.cpp:10,21).
 *   - A batch of `nb_problems` independent problems is stored in one of two layouts:
 *       TRAJOPT_LAYOUT_PROBLEM_MAJOR  X(p, i, j, t) = X[p*rows*cols*slices + i + rows*j + rows*cols*t]
 *                                     (a stack of reference-layout problems; nb_problems == 1 IS the
 *                                      reference layout)
 *       TRAJOPT_LAYOUT_PROBLEM_MINOR  X(p, i, j, t) = X[(i + rows*j + rows*cols*t)*nb_problems + p]
 *                                     (struct-of-arrays, problem index fastest: consecutive lanes read
 *                                      consecutive addresses)
 *     Per-problem scalars (lmbda, diverge, returns ...) are plain [nb_problems] vectors in both layouts.
 *   - Pointers named *_host are host memory; every other data pointer is DEVICE memory of the current
 *     HIP device.  `stream` is a hipStream_t passed as void* (NULL = the default stream).  Batched entry
 *     points are asynchronous on `stream`, allocate nothing, and never synchronise.
 *   - Return value: TRAJOPT_OK or a negative TRAJOPT_ERR_*.  A numerically failed problem (non-PD Quu)
 *     is DATA (`diverge[p]`), never an error (util.cpp:145-148).
 */
#ifndef TRAJOPT_AMD_H
#define TRAJOPT_AMD_H

#include <stdint.h>
#include "trajopt_env.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TRAJOPT_ABI_VERSION 5

#define TRAJOPT_OK 0
#define TRAJOPT_ERR_BAD_ARG (-1)      /* null pointer, non-positive size, unknown enum            */
#define TRAJOPT_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels are built for             */
#define TRAJOPT_ERR_HIP (-3)          /* a HIP runtime call failed; see trajopt_last_hip_error()  */
#define TRAJOPT_ERR_NO_DEVICE (-4)    /* no usable GPU: there is no CPU fallback by design        */
#define TRAJOPT_ERR_TOO_LARGE (-5)    /* beyond the static staging of an allocation-free entry    */

#define TRAJOPT_LAYOUT_PROBLEM_MAJOR 0
#define TRAJOPT_LAYOUT_PROBLEM_MINOR 1

#define TRAJOPT_MAX_ALPHAS 16

int trajopt_abi_version(void);
const char* trajopt_error_string(int code);
const char* trajopt_last_hip_error(void);
/* Number of visible HIP devices, or TRAJOPT_ERR_NO_DEVICE. */
int trajopt_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Regularisation (lambda) schedule of iLQR.run, ilqr.py:166-167 (grow) and :207-208 (shrink).
 * ---------------------------------------------------------------------------------------------- */
typedef struct trajopt_lambda_schedule {
    double min_lmbda;   /* ilqr.py:16  default 1e-6 */
    double max_lmbda;   /* ilqr.py:17  default 1e6  */
    double mult_lmbda;  /* ilqr.py:17  default 1.6  */
} trajopt_lambda_schedule_t;

/* ------------------------------------------------------------------------------------------------
 * A1.  Regularised Riccati / DDP backward pass.
 * Replaces `core.backward_pass` (util.cpp:83-182); called from iLQR.backward_pass (ilqr.py:104-117).
 * ---------------------------------------------------------------------------------------------- */

/* Exact counterpart of the pybind11 entry (util.cpp:83-87,179-180): ONE problem, HOST arrays in the
 * reference layout, scalar lmbda.  Outputs are caller-allocated:
 *   Qxx(n,n,T) Qux(m,n,T) Quu(m,m,T) qx(n,T) qu(m,T) V(n,n,T+1) v(n,T+1) dV(2) K(m,n,T) kff(m,T).
 * Slices the recursion never reaches after a divergence are written as zeros (the reference leaves
 * them uninitialised).  *diverge receives the reference's `_diverge` (time index of the failing step,
 * 0 when none - util.cpp:120,146); *failed (may be NULL) receives the unambiguous flag.
 * Copies to the device, runs the same kernels as the batched entry, copies back, synchronises.  The device staging and
 * its pinned host mirror are kept per device and grown on demand (one copy in, one kernel, one copy out per call);
 * calls on one device are serialised by the library. */
int trajopt_ilqr_backward_pass_host(const double* Cxx_host, const double* cx_host, const double* Cuu_host,
                                    const double* cu_host, const double* Cxu_host, const double* A_host,
                                    const double* B_host, double lmbda, int reg, int dm_state, int dm_act,
                                    int nb_steps, double* Qxx_host, double* Qux_host, double* Quu_host,
                                    double* qx_host, double* qu_host, double* V_host, double* v_host,
                                    double* dV_host, double* K_host, double* kff_host, int* diverge,
                                    int* failed);

/* Allocation-free variant of trajopt_ilqr_backward_pass_host.  It has NO counterpart in the reference: it is for
 * callers that cannot allocate device memory - tight serving / MPC loops, harnesses that forbid hipMalloc (NOT stream
 * capture: the call copies from / to pageable host memory and synchronises the stream, both illegal while capturing).  Same arguments, results and kernels as trajopt_ilqr_backward_pass_host, plus the stream
 * the copies and kernels run on (NULL = the default stream; the call synchronises that stream before returning).
 * The device staging (inputs | outputs | lmbda, diverge) lives in a static device array of the library of
 * TRAJOPT_NOALLOC_MAX_DOUBLES doubles (4 MiB, held per GPU until the process ends) instead of hipMalloc/hipFree:
 * the call performs no allocation of any kind.  A problem needing more staging than that
 * (trajopt_ilqr_backward_pass_host_noalloc_doubles(n, m, T) > TRAJOPT_NOALLOC_MAX_DOUBLES; the limit covers
 * BASELINE's C1-C4 shapes at T = 500, e.g. 12 x 4 x 500 needs 424 384) is rejected with TRAJOPT_ERR_TOO_LARGE before any
 * device work.  Every call on one GPU shares that GPU's array: calls that overlap on one GPU (two host threads, or
 * two streams) race - serialise them. */
#define TRAJOPT_NOALLOC_MAX_DOUBLES 524288
int64_t trajopt_ilqr_backward_pass_host_noalloc_doubles(int dm_state, int dm_act, int nb_steps);
int trajopt_ilqr_backward_pass_host_noalloc(const double* Cxx_host, const double* cx_host, const double* Cuu_host,
                                            const double* cu_host, const double* Cxu_host, const double* A_host,
                                            const double* B_host, double lmbda, int reg, int dm_state, int dm_act,
                                            int nb_steps, double* Qxx_host, double* Qux_host, double* Quu_host,
                                            double* qx_host, double* qu_host, double* V_host, double* v_host,
                                            double* dV_host, double* K_host, double* kff_host, int* diverge,
                                            int* failed, void* stream);

typedef struct trajopt_backward_io {
    /* inputs (ilqr.py:112-114): cost arrays have nb_steps+1 slices, A/B have nb_steps */
    const double* Cxx; const double* cx; const double* Cuu; const double* cu; const double* Cxu;
    const double* A; const double* B;
    /* per-problem regularisation, [nb_problems].  Read; written back when `retry` != 0. */
    double* lmbda;
    double* dlmbda;       /* [nb_problems]; only used (read+written) when retry != 0, else may be NULL */
    /* optional outputs: a NULL pointer skips the store ("gains-only" when all seven are NULL) */
    double* Qxx; double* Qux; double* Quu; double* qx; double* qu; double* V; double* v;
    /* required outputs */
    double* dV;           /* (2,) per problem */
    double* K; double* kff;
    int32_t* diverge;     /* [nb_problems]: -1 = every step passed the PD test, else the failing time index */
    /* optional: uref (m, T) per problem; when non-NULL the kernel also returns
     * g_norm[p] = mean_j max_t |kff_jt| / (|uref_jt| + 1)  (ilqr.py:176) */
    const double* uref;
    double* g_norm;       /* [nb_problems], may be NULL */
    /* optional activity mask [nb_problems]: problems with active[p] == 0 are skipped entirely */
    const int32_t* active;
    /* Optional structured cost.  Every env of the hot path has identity features, so its cost Hessians are
     * closed-form (ilqr/objects.py:74-78 applied to e.g. cartpole.py:112-127): Cxx[t] = w[t] diag(gxx),
     * Cuu[t] = diag(guu), Cxu = 0, the same for every problem.  With cost_w non-NULL the kernels take them from
     * here - w is the [nb_steps+1] weighting vector shared by all problems, gxx = 2 gw (times dt for the LQR
     * envs), guu = 2 uw - and the dense Cxx, Cuu, Cxu pointers above are ignored (may be NULL): 21 of the 51
     * doubles a cartpole step reads.  Supported for PROBLEM_MINOR, dm_state <= 6 (see
     * trajopt_backward_supports_structured_cost); other shapes return TRAJOPT_ERR_UNSUPPORTED. */
    const double* cost_w;
    double cost_gxx[TRAJOPT_ENV_MAX_STATE];
    double cost_guu[TRAJOPT_ENV_MAX_ACT];
    /* Kernel family (ABI v4; per call, so the library keeps no mutable global state): TRAJOPT_KERNEL_AUTO = the
     * library's choice for the shape, layout and batch size; any other value forces that family (the parity tests
     * run every family against the oracle) and returns TRAJOPT_ERR_UNSUPPORTED where it has no kernel for the shape. */
    int32_t kernel;
    int32_t pad_;
    /* Optional multi-buffered gains (ABI v5; the batched driver's, NULL for everybody else).  With out_bank non-NULL,
     * K and kff hold TRAJOPT_GAIN_BANKS banks each and problem p's gains are written to bank out_bank[p]:
     *   PROBLEM_MINOR  K(p, e, t, b) = K[((TRAJOPT_GAIN_BANKS t + b) m n + e) nb_problems + p]   (banks interleaved per time step)
     *   PROBLEM_MAJOR  K(p, e, t, b) = K[(b nb_problems + p) m n T + m n t + e]                   (bank slowest)
     * and the same for kff with m in place of m n.  The other banks of a problem are left untouched: one of them may
     * hold the gains of its last ACCEPTED iterate while a new candidate is computed (ilqr.py:214 without a copy). */
    const int32_t* out_bank;
} trajopt_backward_io_t;

#define TRAJOPT_GAIN_BANKS 3

#define TRAJOPT_KERNEL_AUTO 0
#define TRAJOPT_KERNEL_WORKGROUP 1   /* workgroup = problem, any dx <= 64, du <= 16 (bwd_wg.hip)                    */
#define TRAJOPT_KERNEL_MFMA16 2      /* wave = problem, fp64 matrix cores, dx + du <= 16, du <= 4 (bwd_mfma16.hip)  */
#define TRAJOPT_KERNEL_LANE 3        /* lane = problem, PROBLEM_MINOR, dx <= 6 (bwd_tpp.hip)                        */
#define TRAJOPT_KERNEL_MFMA4 4       /* four problems per wave, PROBLEM_MINOR, dx <= 4, du = 1 (bwd_mfma4.hip)      */
#define TRAJOPT_KERNEL_COLUMNS 5     /* eight lanes per problem, PROBLEM_MINOR, dx = 6, du = 1 (bwd_col8.hip)       */

/* 1 when the batched backward pass accepts io->cost_w for this shape, else 0. */
int trajopt_backward_supports_structured_cost(int dm_state, int dm_act, int nb_steps, int64_t nb_problems, int layout);

/* Batched backward pass on device arrays.  reg: 1 = lmbda*I on Quu, 2 = lmbda*I on V' (util.cpp:135-143),
 * anything else = no regularisation term (as the reference).  retry != 0 repeats a diverged problem
 * inside the kernel with lmbda grown per ilqr.py:166-167 until it passes or lmbda > max_lmbda
 * (ilqr.py:162-173); `sched` may be NULL when retry == 0. */
int trajopt_ilqr_backward_pass_batched(const trajopt_backward_io_t* io, int reg, int dm_state, int dm_act,
                                       int nb_steps, int64_t nb_problems, int layout, int retry,
                                       const trajopt_lambda_schedule_t* sched, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A9.  Time-varying LQR backward sweep with affine dynamics term c ("no-1/2" convention).
 * Replaces the NumPy loop Riccati.backward_pass (riccati.py:100-126).  V = Qxx + Qux^T K uses the
 * matrix product (the reference's elementwise `*` at :123 coincides with it only for dm_act == 1).
 * Outputs: K(m,n,T) kff(m,T) V(n,n,T+1) v(n,T+1).  status[p] (may be NULL): 0 ok, t+1 if Quu was
 * singular at step t (np.linalg.inv would raise).
 * Concurrency of the batched entry: with status non-NULL a call touches only memory its caller passed in (the
 * matrix-core sweep hands the problems it cannot take - asymmetric Cxx slice, non-positive pivot - to the generic
 * kernel through status[p] itself), so calls on different streams or threads of one GPU may overlap freely.  With
 * status == NULL that hand-over uses one static flag array per GPU: at most one such call per GPU at a time.
 * ---------------------------------------------------------------------------------------------- */
int trajopt_riccati_backward_pass_host(const double* Cxx_host, const double* cx_host, const double* Cuu_host,
                                       const double* cu_host, const double* Cxu_host, const double* A_host,
                                       const double* B_host, const double* c_host, int dm_state, int dm_act,
                                       int nb_steps, double* V_host, double* v_host, double* K_host,
                                       double* kff_host, int* status);

int trajopt_riccati_backward_pass_batched(const double* Cxx, const double* cx, const double* Cuu,
                                          const double* cu, const double* Cxu, const double* A, const double* B,
                                          const double* c, int dm_state, int dm_act, int nb_steps,
                                          int64_t nb_problems, int layout, double* V, double* v, double* K,
                                          double* kff, int32_t* status, void* stream);

/* Closed-loop episodes under the sweep's controller: Riccati.forward_pass / .rollout (riccati.py:52-98) with the
 * absolute law u_t = kff_t + K_t x_t (riccati/objects.py:128-129) on the env's step (dynamics + noise, e.g.
 * pendulum.py:131-138).  K (m,n,T), kff (m,T): ONE controller in the reference layout, device memory, shared by all
 * episodes.  x0 (n, N), noise (T, n, N; the caller's draws, NULL = none), state (T+1, n, N), action (T, m, N), xnext
 * (T, n, N; may be NULL) and cost (T+1, N) are PROBLEM_MINOR over the N episodes.  cost_t is taken on the unclipped
 * action with u_{t-1} (zero at t = 0) as the reference does; cost_T = env.cost(x_T, 0, 0, w_T). */
int trajopt_riccati_forward_pass_batched(const trajopt_env_t* env, int nb_steps, int64_t nb_episodes, const double* x0,
                                         const double* K, const double* kff, const double* noise,
                                         const double* weighting, double* state, double* action, double* xnext,
                                         double* cost, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SURVEY 8f.2 (first "next" row after the iLQR path).  Max-entropy (GPS) backward pass:
 * replaces trajopt.gps.core.backward_pass, /root/reference/trajopt/gps/src/util.cpp:272-374 (called from
 * gps/mbgps.py:250-260): Q.. = -(C.. + F'V'F)/alpha with the affine dynamics term c, the process noise sigma_dyn
 * (n,n,T), the constant c0 (T+1) and the per-step temperature alpha (T); outputs as the reference's tuple
 * Qxx Qux Quu qx qu q0 V v v0 K kff sigma_ctl.  Shapes with a kernel: trajopt_gps_backward_supported() - any
 * dm_state <= 32, dm_act <= 16 (lane-per-problem kernels for the reference's own small shapes, a wave-per-problem
 * kernel with run-time dimensions for the rest).
 * ---------------------------------------------------------------------------------------------- */
typedef struct trajopt_gps_backward_io {
    const double* Cxx; const double* cx; const double* Cuu; const double* cu; const double* Cxu; const double* c0;
    const double* A; const double* B; const double* c; const double* sigma_dyn;
    const double* alpha;   /* (T,) per problem */
    /* optional outputs (NULL skips the store) */
    double* Qxx; double* Qux; double* Quu; double* qx; double* qu; double* q0;
    /* required outputs */
    double* V; double* v; double* v0; double* K; double* kff; double* sigma_ctl;
    int32_t* diverge;      /* [nb_problems]: -1 = the sweep completed, else the time index it stopped at */
    int32_t* status;       /* [nb_problems], may be NULL: 0 ok, 1 Quu positive definite (util.cpp:333-336),
                              2 Quu singular (Armadillo's inv() throws) */
} trajopt_gps_backward_io_t;

int trajopt_gps_backward_supported(int dm_state, int dm_act);

/* device arrays, PROBLEM_MINOR (or one problem in either layout) */
int trajopt_gps_backward_pass_batched(const trajopt_gps_backward_io_t* io, int dm_state, int dm_act, int nb_steps,
                                      int64_t nb_problems, int layout, void* stream);

/* host arrays in the reference's layout; *diverge as the reference's int (0 = none) */
int trajopt_gps_backward_pass_host(const double* Cxx_host, const double* cx_host, const double* Cuu_host,
                                   const double* cu_host, const double* Cxu_host, const double* c0_host,
                                   const double* A_host, const double* B_host, const double* c_host,
                                   const double* sigma_dyn_host, const double* alpha_host, int dm_state, int dm_act,
                                   int nb_steps, double* Qxx_host, double* Qux_host, double* Quu_host, double* qx_host,
                                   double* qu_host, double* q0_host, double* V_host, double* v_host, double* v0_host,
                                   double* K_host, double* kff_host, double* sigma_ctl_host, int* diverge,
                                   int* status);

/* The other functions of trajopt.gps.core, same layouts and shape set (PROBLEM_MINOR, or one problem):
 *   forward_pass      /root/reference/trajopt/gps/src/util.cpp:195-269  (gps/mbgps.py:140-145)  Gaussian moment propagation;
 *                     the parts of the last mu_xu / sigma_xu slices the reference leaves unset are written as zero
 *   augment_cost      util.cpp:136-193 (mbgps.py:166-170)   cost + alpha * log-policy terms; T+1 slices, the last copied
 *   kl_divergence     util.cpp:83-121  (mbgps.py:198-201)   per-step KL(p || q) of two linear-Gaussian controllers
 *   quad_expectation  util.cpp:123-134 (mbgps.py:184-186)   E[x'Qx + x'q + q0] under N(mu, sigma)
 * status[p] = 1 when a sigma_ctl is not positive definite (Armadillo's inv_sympd() throws there), else 0.
 * augment_cost and kl_divergence launch one grid row per step: nb_steps < 65535 (TRAJOPT_ERR_BAD_ARG beyond). */
/* The four other functions of the module take the same shapes as the backward pass (any dm_state <= 32, dm_act <= 16). */
int trajopt_gps_forward_pass_batched(const double* mu_x0, const double* sigma_x0, const double* A, const double* B,
                                     const double* c, const double* sigma_dyn, const double* K, const double* kff,
                                     const double* sigma_ctl, int dm_state, int dm_act, int nb_steps,
                                     int64_t nb_problems, int layout, double* mu_x, double* sigma_x, double* mu_u,
                                     double* sigma_u, double* mu_xu, double* sigma_xu, void* stream);
int trajopt_gps_augment_cost_batched(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                     const double* Cxu, const double* c0, const double* K, const double* kff,
                                     const double* sigma_ctl, const double* alpha, int dm_state, int dm_act,
                                     int nb_steps, int64_t nb_problems, int layout, double* agCxx, double* agcx,
                                     double* agCuu, double* agcu, double* agCxu, double* agc0, int32_t* status,
                                     void* stream);
int trajopt_gps_kl_divergence_batched(const double* p_K, const double* p_kff, const double* p_sigma_ctl,
                                      const double* q_K, const double* q_kff, const double* q_sigma_ctl,
                                      const double* mu_x, const double* sigma_x, int dm_state, int dm_act,
                                      int nb_steps, int64_t nb_problems, int layout, double* kl, int32_t* status,
                                      void* stream);
int trajopt_gps_quad_expectation_batched(const double* mu, const double* sigma_s, const double* Q, const double* q,
                                         const double* q0, int dm_state, int64_t nb_problems, int layout,
                                         double* result, void* stream);
/* host arrays in the reference's layout, one problem */
int trajopt_gps_forward_pass_host(const double* mu_x0_host, const double* sigma_x0_host, const double* A_host,
                                  const double* B_host, const double* c_host, const double* sigma_dyn_host,
                                  const double* K_host, const double* kff_host, const double* sigma_ctl_host,
                                  int dm_state, int dm_act, int nb_steps, double* mu_x_host, double* sigma_x_host,
                                  double* mu_u_host, double* sigma_u_host, double* mu_xu_host, double* sigma_xu_host);
int trajopt_gps_augment_cost_host(const double* Cxx_host, const double* cx_host, const double* Cuu_host,
                                  const double* cu_host, const double* Cxu_host, const double* c0_host,
                                  const double* K_host, const double* kff_host, const double* sigma_ctl_host,
                                  const double* alpha_host, int dm_state, int dm_act, int nb_steps,
                                  double* agCxx_host, double* agcx_host, double* agCuu_host, double* agcu_host,
                                  double* agCxu_host, double* agc0_host, int* status);
int trajopt_gps_kl_divergence_host(const double* p_K_host, const double* p_kff_host, const double* p_sigma_ctl_host,
                                   const double* q_K_host, const double* q_kff_host, const double* q_sigma_ctl_host,
                                   const double* mu_x_host, const double* sigma_x_host, int dm_state, int dm_act,
                                   int nb_steps, double* kl_host, int* status);
int trajopt_gps_quad_expectation_host(const double* mu_host, const double* sigma_s_host, const double* Q_host,
                                      const double* q_host, double q0, int dm_state, double* result);

/* ------------------------------------------------------------------------------------------------
 * SURVEY 8f.4.  Belief-space iLQR backward pass: replaces trajopt.bspilqr.core.backward_pass,
 * /root/reference/trajopt/bspilqr/src/util.cpp:83-204 (called at bspilqr/bspilqr.py:90-104).  b = dm_belief; the
 * covariance channel has b*b entries: p, tau (b*b, .), T, X (b*b, b, .), U, Y (b*b, b*b, .), V, Z (b*b, dm_act, .).
 * Q, q, p are read up to slice nb_steps, everything else up to nb_steps - 1.  dS starts at zero (uninitialised in
 * the reference, :132).  tau_mode: what util.cpp:158's read of the not-yet-assigned tau.col(i) means -
 * 0 = tau.col(i+1) (the evident intent, as :152,:155), 1 = zeros.  Shapes: any b <= 8, dm_act <= 8 (lane-per-problem
 * kernels for b in {2, 4}, dm_act in {1, 2}; a wave-per-problem kernel with run-time dimensions for the rest).
 * ---------------------------------------------------------------------------------------------- */
typedef struct trajopt_bsp_backward_io {
    const double* Q; const double* q; const double* R; const double* r; const double* P; const double* p;
    const double* F; const double* G; const double* T; const double* U; const double* V;
    const double* X; const double* Y; const double* Z;
    const double* lmbda;   /* [nb_problems] */
    double* S; double* s; double* tau;
    double* dS;            /* (2,) per problem */
    double* K; double* kff;
    int32_t* diverge;      /* [nb_problems]: -1 = completed, else the failing time index (:170-173) */
} trajopt_bsp_backward_io_t;

int trajopt_bspilqr_backward_supported(int dm_belief, int dm_act);
int trajopt_bspilqr_backward_pass_batched(const trajopt_bsp_backward_io_t* io, int reg, int tau_mode, int dm_belief,
                                          int dm_act, int nb_steps, int64_t nb_problems, int layout, void* stream);
int trajopt_bspilqr_backward_pass_host(const double* Q_host, const double* q_host, const double* R_host,
                                       const double* r_host, const double* P_host, const double* p_host,
                                       const double* F_host, const double* G_host, const double* T_host,
                                       const double* U_host, const double* V_host, const double* X_host,
                                       const double* Y_host, const double* Z_host, double lmbda, int reg,
                                       int tau_mode, int dm_belief, int dm_act, int nb_steps, double* S_host,
                                       double* s_host, double* tau_host, double* dS_host, double* K_host,
                                       double* kff_host, int* diverge, int* failed);

/* ------------------------------------------------------------------------------------------------
 * A8.  Env model evaluated on the device: xn = env.dynamics(x, u), c = env.cost(x, u, u_last, a)
 * (e.g. cartpole.py:62-97,112-127).  x (n, N), u (m, N), PROBLEM_MINOR or MAJOR per `layout`.
 * Either output may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int trajopt_env_step_batched(const trajopt_env_t* env, const double* x, const double* u, const double* u_last,
                             const double* a, int64_t nb_problems, int layout, double* xn, double* cost,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * A3.  Forward rollout under the affine law, for every (problem, alpha) pair:
 *   u_t = clip(uref_t + alpha*kff_t + K_t (x_t - xref_t), +-umax); cost_t = env.cost(x_t, u_t, u_t, w_t);
 *   x_{t+1} = env.dynamics(x_t, u_t); cost_T = env.cost(x_T, 0, 0, w_T)         (ilqr.py:89-102,
 *   objects.py:137-139).
 * x0 (n,) per problem.  alphas_host[nb_alphas] (<= TRAJOPT_MAX_ALPHAS).  weighting [nb_steps+1] device.
 * Candidate outputs carry a leading alpha axis: state[a] (n, T+1), action[a] (m, T), cost[a] (T+1)
 * per problem in `layout` (alpha slowest); any of the three may be NULL.
 * ret[a*nb_problems + p] = sum_t cost;  finite[a*nb_problems + p] = all(state < 1e8) (ilqr.py:141).
 * ---------------------------------------------------------------------------------------------- */
int trajopt_ilqr_forward_pass_batched(const trajopt_env_t* env, int nb_steps, int64_t nb_problems, int layout,
                                      const double* x0, const double* K, const double* kff, const double* xref,
                                      const double* uref, const double* weighting, const double* alphas_host,
                                      int nb_alphas, const int32_t* active, double* state, double* action,
                                      double* cost, double* ret, int32_t* finite, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A4 + A5 (+A10).  Quadratisation along a reference trajectory.
 *   A_t = d dynamics/dx, B_t = d dynamics/du at (xref_t, uref_t), t < T          (objects.py:114-117)
 *   Cxx, cx, Cuu, cu, Cxu at (xref_t, [uref,0]_t, [uref,0]_{t-1}, w_t), t <= T   (objects.py:68-78)
 * convention 0 = iLQR (no 1/2, no de-centring); 1 = Riccati (riccati/objects.py:54-65,102-108: Hessians
 * halved, linear terms de-centred, affine residual c (n, T) also written; `c` must be non-NULL).
 * Derivatives are exact (forward-mode dual numbers through clip -> RK4 -> clip, autograd's clip rule).
 * ---------------------------------------------------------------------------------------------- */
int trajopt_quadratize_batched(const trajopt_env_t* env, int nb_steps, int64_t nb_problems, int layout,
                               int convention, const double* xref, const double* uref, const double* weighting,
                               const int32_t* active, double* A, double* B, double* c, double* Cxx, double* cx,
                               double* Cuu, double* cu, double* Cxu, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A6.  Batched iLQR driver: everything iLQR.run keeps per problem (ilqr.py:136-237), resident on the
 * device, PROBLEM_MINOR layout except for the gains group (see gains_layout).  The caller (PyTorch on the
 * Python side) owns every buffer.
 * ---------------------------------------------------------------------------------------------- */
typedef struct trajopt_ilqr_options {
    trajopt_lambda_schedule_t sched;
    double tolfun;    /* ilqr.py:17 */
    double tolgrad;   /* ilqr.py:18 */
    double min_imp;   /* ilqr.py:18 */
    int32_t reg;      /* ilqr.py:18 */
    int32_t nb_alphas;
    double alphas[TRAJOPT_MAX_ALPHAS];  /* ilqr.py:15 */
    int32_t argmin_alpha;  /* 0 = reference's first-accept order (ilqr.py:187-202); 1 = lowest return among accepted */
    int32_t keep_value;    /* 1 = also maintain vfunc/qfunc (V, v, Qxx, Qux, Quu, qx, qu) on accept    */
    /* Line-search scheduling (results are identical either way):
     *   0 = every alpha of every problem in one launch (shortest latency: small batches);
     *   1 = staged like the reference's sequential loop (ilqr.py:187-202): alphas[0] for everybody, then the
     *       remaining alphas only for the problems that rejected it, compacted so that waves stay full (needs
     *       ws->pending_list / ws->pending_count; ignored with argmin_alpha);
     *   2 = two passes: the returns of every (problem, alpha) without writing any trajectory, then the accepted alpha
     *       of each problem rolled out once more and written (large batches, where the stores of the candidates nobody
     *       reads dominate; built-in models with the plain cost, otherwise 0 is taken);
     *  -1 = pick by batch size. */
    int32_t staged_line_search;
    /* 1 = never fuse the quadratisation into the backward kernel (fused_qb4.hip): always the two launches
     * quadratise -> backward pass.  Same results to round-off; for tests and measurements. */
    int32_t no_fusion;
} trajopt_ilqr_options_t;

typedef struct trajopt_ilqr_workspace {
    int32_t nb_steps;
    /* Layout of the "gains group": A, B, Cxx, cx, Cuu, cu, Cxu, K, kff, cand_K, cand_kff and the vfunc/qfunc
     * arrays.  Everything else (x0, xref, uref, candidates, per-problem scalars) is always PROBLEM_MINOR.
     * PROBLEM_MINOR suits the lane-per-problem backward kernel (dx <= 6), PROBLEM_MAJOR the matrix-core one. */
    int32_t gains_layout;
    int64_t nb_problems;
    /* problem data */
    const double* x0;         /* (n,)            */
    const double* weighting;  /* [T+1], shared   */
    /* accepted solution (ilqr.py:210-217) */
    double* xref; double* uref;          /* (n,T+1), (m,T)          */
    double* K; double* kff;              /* ctl: (m,n,T), (m,T)      */
    double* V; double* v;                /* vfunc, only if keep_value */
    double* Qxx; double* Qux; double* Quu; double* qx; double* qu;   /* qfunc, only if keep_value */
    /* per-problem scalars, [nb_problems] */
    double* last_return; double* lmbda; double* dlmbda; double* alpha;
    int32_t* status;      /* 0 running, 1 tolgrad, 2 tolfun, 3 lmbda > max (ilqr.py:177,222,168/228)  */
    int32_t* iters;       /* iterations executed                                                      */
    int32_t* trace_len;   /* entries written to trace                                                 */
    double* trace;        /* [trace_cap][nb_problems] accepted returns (ilqr.py:149,219)             */
    int32_t trace_cap; int32_t pad2_;
    /* scratch */
    double* A; double* B; double* Cxx; double* cx; double* Cuu; double* cu; double* Cxu;   /* quadratisation */
    double* cand_K; double* cand_kff;    /* backward-pass candidate gains                             */
    double* cand_V; double* cand_v; double* cand_Qxx; double* cand_Qux; double* cand_Quu; double* cand_qx;
    double* cand_qu;                     /* only if keep_value                                        */
    double* dV; double* g_norm;
    int32_t* diverge; int32_t* active;
    int32_t* sel;                        /* [nb_problems] index of the accepted alpha, -1 = none       */
    double* cand_state; double* cand_action; double* cand_ret;   /* [nb_alphas] candidates            */
    int32_t* cand_finite;
    int32_t* pending_list;   /* [nb_problems], staged line search only (may be NULL otherwise) */
    int32_t* pending_count;  /* [1] */
    /* ABI v5: no copy on accept.  cand_state / cand_action hold 2 * nb_alphas + 1 planes - two banks of nb_alphas and
     * one plane nobody reads (lanes of stopped problems write there).  Iteration i of a trajopt_ilqr_iterate call
     * writes its candidates into bank i % 2, for every problem, and accepting alpha k (ilqr.py:210-217) is
     *     ref_plane[p] = (i % 2) * nb_alphas + k,   gains_bank[p] = out_bank[p]:
     * the reference trajectory of problem p is then read from that plane.  A reference still in the bank about to be
     * written (its problem has rejected every alpha since) already has a copy in xref / uref - the quadratisation
     * writes what it reads through, every iteration - so ref_plane[p] just becomes -1 = "in xref / uref".
     * cand_K / cand_kff hold TRAJOPT_GAIN_BANKS banks (layout: trajopt_backward_io_t::out_bank): problem p writes its
     * gains to bank out_bank[p] = i % 2, unless that bank holds its accepted controller (gains_bank[p]; it has rejected
     * every alpha since) - then to bank 2.  Nothing is ever copied, and all but the rejecting problems write the same
     * bank, whole cache lines at a time.  gains_bank[p] = -1 means "in K / kff".
     * At the end of every call xref / uref / K / kff are brought up to date and ref_plane / gains_bank are -1 again -
     * the state a fresh workspace must have.
     * Neighbouring problems read different planes, so a plane is TIME-BLOCKED - 8 consecutive steps of one element
     * of one problem are one 64-byte run, which its reader uses in full:
     *     plane(p, e, t) = plane[(((t / 8) rows + e) nb_problems + p) 8 + t % 8],
     *     plane size = ceil(slices / 8) * 8 * rows * nb_problems   (state: rows n, slices T + 1; action: m, T). */
    int32_t* ref_plane;      /* [nb_problems] */
    int32_t* gains_bank;     /* [nb_problems] */
    int32_t* out_bank;       /* [nb_problems] */
} trajopt_ilqr_workspace_t;

/* ilqr.py:139-149: initial rollout of ws->K / ws->kff over the alphas; first one with all(state < 1e8)
 * becomes xref/uref/last_return and trace[0]; ref_plane / gains_bank are reset to -1. */
int trajopt_ilqr_init_rollout(const trajopt_env_t* env, const trajopt_ilqr_options_t* opt,
                              const trajopt_ilqr_workspace_t* ws, void* stream);
/* `nb_iter` iterations of ilqr.py:151-231 for every problem whose status is 0, without any host
 * synchronisation: quadratise -> backward pass with in-kernel lambda retries -> gradient-norm stop ->
 * rollout over all alphas -> first-accept selection -> accept/reject and lambda update.
 * The call keeps no state outside `ws`: calls on DIFFERENT workspaces may overlap on different streams of one GPU (while no
 * trajopt_profile_begin is open).  Problems are independent, so a batch can be cut into a few workspaces that iterate side
 * by side - one's rollout under another's backward pass; measured gains and the rule the Python host side uses:
 * trajopt_amd/ilqr/streamed.py. */
int trajopt_ilqr_iterate(const trajopt_env_t* env, const trajopt_ilqr_options_t* opt,
                         const trajopt_ilqr_workspace_t* ws, int nb_iter, void* stream);

/* What trajopt_ilqr_iterate will launch for this env / options / workspace (no counterpart in the reference; used by
 * bench.py to attribute bytes and flops to the kernels that actually run). */
typedef struct trajopt_ilqr_plan {
    int32_t structured_cost;   /* 1: closed-form Cxx / Cuu / Cxu formed inside the backward kernel (cost_w)            */
    int32_t fused_quad_bwd;    /* quadratisation and backward pass in one launch: 1 = four problems per wave (fused_qb4.hip,
                                  < 32768 problems), 2 = lane per problem (bwd_tpp.hip, >= 32768 problems); 0 = two launches */
    int32_t staged_line_search;/* 0 one launch; 1: alphas[0] for everybody, the rest for the compacted list of rejecting
                                  problems; 2: returns of every alpha first, then the accepted one again, written     */
    int32_t backward_kernel;   /* TRAJOPT_KERNEL_* family of the backward pass                                         */
    int32_t rollout_straight_line; /* 1: the rollout's lone-wave form (8 steps of straight-line code per block)        */
    int32_t pad_;
} trajopt_ilqr_plan_t;
int trajopt_ilqr_plan(const trajopt_env_t* env, const trajopt_ilqr_options_t* opt, const trajopt_ilqr_workspace_t* ws,
                      trajopt_ilqr_plan_t* plan);

/* Per-phase device timing of trajopt_ilqr_iterate (measurement aid; no counterpart in the reference, which
 * has no timers).  begin() creates hipEvents for up to `max_iterations` iterations; while enabled, iterate
 * records them on ITS launch stream around the six phases  prep | quadratise | backward | post-backward |
 * rollout | select+commit.  end() waits for the last event, writes the summed milliseconds per phase to
 * phase_ms[6] and the number of recorded iterations, and disables recording.  Not thread-safe. */
#define TRAJOPT_PROFILE_PHASES 6
int trajopt_profile_begin(int max_iterations);
int trajopt_profile_end(double* phase_ms, int* nb_iterations);
/* after trajopt_profile_begin: bracket only the phases in phase_mask (bit k = phase k; others report 0) - an event costs
 * the stream about 5 us, so a timed region should carry only the ones it needs */
int trajopt_profile_select(unsigned phase_mask);

/* Layout conversion helper (device to device): dst = src re-laid-out; rows*cols*slices elements/problem. */
int trajopt_convert_layout(const double* src, double* dst, int64_t elems_per_problem, int64_t nb_problems,
                           int src_layout, int dst_layout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRAJOPT_AMD_H */
