// extern "C" entry points of libtrajopt_amd.so (declared in include/trajopt_amd.h) for the backward
// passes, plus the small utilities.  No CPU fallback exists anywhere: without a HIP device every
// compute entry point returns TRAJOPT_ERR_NO_DEVICE / TRAJOPT_ERR_HIP.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "common.h"

namespace trajopt {

static thread_local char g_hip_err[512] = "";
void set_hip_error(hipError_t e, const char* where) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}


template <bool TO_MINOR>
__global__ void convert_layout_kernel(const double* __restrict__ src, double* __restrict__ dst, size_t E, size_t NP) {
    // grid-stride over all E*NP elements; index by destination so stores are coalesced
    const size_t total = E * NP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        if (TO_MINOR) {  // dst[e*NP + p] = src[p*E + e]
            const size_t e = i / NP, p = i % NP;
            dst[i] = src[p * E + e];
        } else {         // dst[p*E + e] = src[e*NP + p]
            const size_t p = i / E, e = i % E;
            dst[i] = src[e * NP + p];
        }
    }
}

// RAII device buffer for the *_host convenience wrappers only (the batched API never allocates).
struct DevBuf {
    double* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { n = count; return hipMalloc(&p, (count ? count : 1) * sizeof(double)); }
};

// Above this batch size the lane-per-problem kernel (full waves, every access a whole 512-byte row) wins over the
// four-problems-per-wave matrix-core kernel (shortest dependent chain per step, but a wave uses only 32 bytes of each
// 128-byte line it touches, so it saturates the texture addressers at ~5.4 G problem-steps/s).  Measured crossover
// on MI355X, dx = 4: 0.37 vs 0.57 ms at 4096 problems, 0.76 vs 0.60 ms at 8192.
constexpr int64_t kMfma4MaxProblems = 6144;

// Up to this batch size the column-per-lane kernel (bwd_col8.hip: eight lanes per problem, a step is a third of the
// dependent instructions but every wave recomputes the shared quantities of its 8 problems) beats lane = problem for
// dx = 6: 8192 problems are 1024 waves instead of 128.  Beyond it the chip is full either way and the redundancy loses.
// Measured (MI355X, T = 500; column kernel / lane kernel): structured cost, whole iLQR iteration - 12 288 problems 5.26 /
// 5.35 ms, 16 384 5.94 / 5.92, 24 576 9.35 / 8.79; dense cost blocks, backward pass alone, gains-only - 8192 1.05 / 1.56 ms,
// 10 240 1.69 / 1.45, 12 288 1.87 / 1.48, 24 576 3.16 / 2.40 (full outputs: 2.07 / 2.30, 3.52 / 2.08, 3.84 / 2.14, 6.34 / 4.42).
constexpr int64_t kCol8MaxProblems = 16384, kCol8MaxProblemsDense = 8192;

static bool tpp_offsets_fit(int n, int64_t NP) { return (double)n * n * (double)NP * 8.0 + 512.0 < 4294967296.0; }

}  // namespace trajopt

namespace trajopt {

// Kernel family of the regularised backward pass for a shape (TRAJOPT_KERNEL_*), or -1 when `want` names a family
// without a kernel for it.  want = TRAJOPT_KERNEL_AUTO:  four problems per wave for small struct-of-arrays batches of
// dx <= 4, du = 1  ->  lane = problem for the struct-of-arrays layout and dx <= 6 (NP == 1: the two layouts are the
// same memory)  ->  wave = problem on the fp64 matrix cores (dx + du <= 16, du <= 4; best with PROBLEM_MAJOR, whose
// per-step slices are contiguous)  ->  workgroup = problem.
int backward_kernel_family(int n, int m, int T, int64_t NP, int layout, int want, bool structured) {
    const bool minor = layout == TRAJOPT_LAYOUT_PROBLEM_MINOR || NP == 1;
    const bool mfma4_ok = minor && m == 1 && n >= 2 && n <= 4 && (double)(T + 2) * n * n * (double)NP * 8.0 < 4294967296.0;
    const bool lane_ok = minor && tpp_offsets_fit(n, NP) &&
                         ((m == 1 && (n == 2 || n == 3 || n == 4 || n == 6)) || (n == 4 && m == 2));
    const bool col8_ok = minor && col8_shape_ok(n, m, T, NP);
    const bool mfma16_ok = n + m <= 16 && m <= 4 && n >= 1;
    const bool wg_ok = n <= 64 && m <= 16;
    switch (want) {
        case TRAJOPT_KERNEL_AUTO:
            if (mfma4_ok && NP <= kMfma4MaxProblems) return TRAJOPT_KERNEL_MFMA4;
            if (col8_ok && NP <= (structured ? kCol8MaxProblems : kCol8MaxProblemsDense)) return TRAJOPT_KERNEL_COLUMNS;
            if (lane_ok) return TRAJOPT_KERNEL_LANE;
            if (mfma16_ok) return TRAJOPT_KERNEL_MFMA16;
            return wg_ok ? TRAJOPT_KERNEL_WORKGROUP : -1;
        case TRAJOPT_KERNEL_MFMA4: return mfma4_ok ? want : -1;
        case TRAJOPT_KERNEL_LANE: return lane_ok ? want : -1;
        case TRAJOPT_KERNEL_COLUMNS: return col8_ok ? want : -1;
        case TRAJOPT_KERNEL_MFMA16: return mfma16_ok ? want : -1;
        case TRAJOPT_KERNEL_WORKGROUP: return wg_ok ? want : -1;
        default: return -1;
    }
}

}  // namespace trajopt

using namespace trajopt;

extern "C" {

int trajopt_abi_version(void) { return TRAJOPT_ABI_VERSION; }

const char* trajopt_error_string(int code) {
    switch (code) {
        case TRAJOPT_OK: return "ok";
        case TRAJOPT_ERR_BAD_ARG: return "bad argument";
        case TRAJOPT_ERR_UNSUPPORTED: return "unsupported shape";
        case TRAJOPT_ERR_HIP: return "HIP runtime error";
        case TRAJOPT_ERR_NO_DEVICE: return "no HIP device (there is no CPU fallback)";
        case TRAJOPT_ERR_TOO_LARGE: return "problem exceeds the static device staging of the allocation-free entry point";
        default: return "unknown error";
    }
}

const char* trajopt_last_hip_error(void) { return g_hip_err; }

int trajopt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TRAJOPT_ERR_NO_DEVICE;
    return n;
}

int trajopt_backward_supports_structured_cost(int n, int m, int T, int64_t NP, int layout) {
    // the family the dispatcher below picks by itself for this shape must take it: lane-per-problem, four-problems-
    // per-wave, column-per-lane (struct-of-arrays, dx <= 6) or wave-per-problem on the matrix cores (dx + du <= 16,
    // du <= 4, either layout); the workgroup kernel does not
    if (n <= 0 || m <= 0 || T <= 0 || NP <= 0) return 0;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return 0;
    const int fam = backward_kernel_family(n, m, T, NP, layout, TRAJOPT_KERNEL_AUTO, true);
    return (fam == TRAJOPT_KERNEL_LANE || fam == TRAJOPT_KERNEL_MFMA4 || fam == TRAJOPT_KERNEL_COLUMNS || fam == TRAJOPT_KERNEL_MFMA16) ? 1 : 0;
}

int trajopt_ilqr_backward_pass_batched(const trajopt_backward_io_t* io, int reg, int n, int m, int T, int64_t NP,
                                       int layout, int retry, const trajopt_lambda_schedule_t* sched,
                                       void* stream) {
    if (!io || n <= 0 || m <= 0 || T <= 0 || NP <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
    if (!io->cx || !io->cu || !io->A || !io->B || !io->lmbda || !io->dV || !io->K || !io->kff || !io->diverge)
        return TRAJOPT_ERR_BAD_ARG;
    if (!io->cost_w && (!io->Cxx || !io->Cuu || !io->Cxu)) return TRAJOPT_ERR_BAD_ARG;
    if (io->cost_w && !trajopt_backward_supports_structured_cost(n, m, T, NP, layout)) return TRAJOPT_ERR_UNSUPPORTED;
    if (retry && !sched) return TRAJOPT_ERR_BAD_ARG;
    trajopt_lambda_schedule_t s = sched ? *sched : trajopt_lambda_schedule_t{1e-6, 1e6, 1.6};
    hipStream_t st = as_stream(stream);
    const int fam = backward_kernel_family(n, m, T, NP, layout, io->kernel, io->cost_w != nullptr);
    if (fam < 0 || (io->cost_w && fam != TRAJOPT_KERNEL_LANE && fam != TRAJOPT_KERNEL_MFMA4 && fam != TRAJOPT_KERNEL_COLUMNS && fam != TRAJOPT_KERNEL_MFMA16)) return TRAJOPT_ERR_UNSUPPORTED;
    bool handled = false;
    int rc = TRAJOPT_OK;
    switch (fam) {
        case TRAJOPT_KERNEL_MFMA4: rc = launch_backward_mfma4(*io, reg, n, m, T, NP, retry, s, st, &handled); break;
        case TRAJOPT_KERNEL_LANE: rc = launch_backward_tpp(*io, reg, n, m, T, NP, retry, s, st, &handled); break;
        case TRAJOPT_KERNEL_COLUMNS: rc = launch_backward_col8(*io, reg, n, m, T, NP, retry, s, st, &handled); break;
        case TRAJOPT_KERNEL_MFMA16: rc = launch_backward_mfma16(*io, reg, n, m, T, NP, layout, retry, s, st, &handled); break;
        default: break;
    }
    if (rc != TRAJOPT_OK || handled) return rc;
    if (fam != TRAJOPT_KERNEL_WORKGROUP) return TRAJOPT_ERR_UNSUPPORTED;  // (the family table and a launcher disagree)
    return launch_backward_wg(*io, reg, n, m, T, NP, layout, retry, s, st);
}

int trajopt_riccati_backward_pass_batched(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                          const double* Cxu, const double* A, const double* B, const double* c,
                                          int n, int m, int T, int64_t NP, int layout, double* V, double* v,
                                          double* K, double* kff, int32_t* status, void* stream) {
    if (!Cxx || !cx || !Cuu || !cu || !Cxu || !A || !B || !c || !V || !v || !K || !kff) return TRAJOPT_ERR_BAD_ARG;
    if (n <= 0 || m <= 0 || T <= 0 || NP <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
    {  // tiled matrix-core kernel (16 < dx <= 64, PROBLEM_MAJOR)
        bool handled = false;
        const int rc = launch_riccati_mfma64(Cxx, cx, Cuu, cu, Cxu, A, B, c, n, m, T, NP, layout, V, v, K, kff, status,
                                             as_stream(stream), &handled);
        if (rc != TRAJOPT_OK || handled) return rc;
    }
    return launch_riccati_wg(Cxx, cx, Cuu, cu, Cxu, A, B, c, n, m, T, NP, layout, V, v, K, kff, status,
                             as_stream(stream));
}

int trajopt_convert_layout(const double* src, double* dst, int64_t E, int64_t NP, int src_layout, int dst_layout,
                           void* stream) {
    if (!src || !dst || E <= 0 || NP <= 0) return TRAJOPT_ERR_BAD_ARG;
    hipStream_t st = as_stream(stream);
    if (src_layout == dst_layout || NP == 1 || E == 1) {
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)E * NP * sizeof(double), hipMemcpyDeviceToDevice, st));
        return TRAJOPT_OK;
    }
    const size_t total = (size_t)E * NP;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dst_layout == TRAJOPT_LAYOUT_PROBLEM_MINOR)
        hipLaunchKernelGGL(convert_layout_kernel<true>, dim3(grid), dim3(256), 0, st, src, dst, (size_t)E, (size_t)NP);
    else
        hipLaunchKernelGGL(convert_layout_kernel<false>, dim3(grid), dim3(256), 0, st, src, dst, (size_t)E, (size_t)NP);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

// ---- host-pointer wrappers: the exact shape of the reference's pybind11 entry ------------------------
}  // extern "C"

namespace trajopt {

// doubles of device staging one host-array backward pass needs: inputs | outputs | lmbda, diverge
static size_t backward_host_staging_doubles(int n, int m, int T) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m, T1 = (size_t)T + 1;
    return (nn + n + mm + m + nm) * T1 + (nn + nm) * (size_t)T               // Cxx cx Cuu cu Cxu | A B
           + (nn + nm + mm + n + m + nm + m) * (size_t)T + (nn + n) * T1 + 2  // Qxx Qux Quu qx qu K kff | V v | dV
           + 2;                                                               // lmbda, diverge
}

// Staging of the *_host convenience entry points: one device buffer and one pinned host mirror per device, grown on
// demand and kept for the process (a call used to pay hipMalloc + hipFree + 19 pageable copies: 0.29 ms for a T = 100
// pendulum; now one copy in, one kernel, one copy out).  The mutex is held for the whole call: host-array calls on one
// device are serialised (the batched entry points are the concurrent interface).
struct HostStage {
    double *dev = nullptr, *pin = nullptr;
    size_t cap = 0;
};
static std::mutex g_stage_mutex;
static HostStage g_stage[64];
static int host_stage_locked(size_t count, HostStage** out) {   // g_stage_mutex held
    int dev = -1;
    TRAJOPT_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return TRAJOPT_ERR_UNSUPPORTED;
    HostStage& h = g_stage[dev];
    if (h.cap < count) {
        if (h.dev) (void)hipFree(h.dev);
        if (h.pin) (void)hipHostFree(h.pin);
        h = HostStage{};
        const size_t cap = count + count / 2 + 1024;
        TRAJOPT_HIP_CHECK(hipMalloc(&h.dev, cap * sizeof(double)));
        if (hipHostMalloc(&h.pin, cap * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            (void)hipFree(h.dev);
            h = HostStage{};
            return TRAJOPT_ERR_HIP;
        }
        h.cap = cap;
    }
    *out = &h;
    return TRAJOPT_OK;
}

// The body shared by trajopt_ilqr_backward_pass_host (cached staging, pinned mirror) and its allocation-free variant
// (staging = the library's static device array, no mirror): copy in, zero the outputs, run the batched kernels, copy out.
static int backward_host_on_staging(double* staging, double* pinned, hipStream_t st, const double* Cxx, const double* cx,
                                    const double* Cuu, const double* cu, const double* Cxu, const double* A,
                                    const double* B, double lmbda, int reg, int n, int m, int T, double* Qxx,
                                    double* Qux, double* Quu, double* qx, double* qu, double* V, double* v,
                                    double* dV, double* K, double* kff, int* diverge, int* failed) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const size_t in_counts[7] = {nn * (T + 1), (size_t)n * (T + 1), mm * (T + 1), (size_t)m * (T + 1), nm * (T + 1),
                                 nn * T, nm * T};
    const double* in_host[7] = {Cxx, cx, Cuu, cu, Cxu, A, B};
    const size_t out_counts[10] = {nn * T, nm * T, mm * T, (size_t)n * T, (size_t)m * T, nn * (T + 1),
                                   (size_t)n * (T + 1), 2, nm * T, (size_t)m * T};
    double* out_host[10] = {Qxx, Qux, Quu, qx, qu, V, v, dV, K, kff};
    size_t in_total = 0;
    for (size_t c : in_counts) in_total += c;
    const size_t total = backward_host_staging_doubles(n, m, T);
    double* d = staging;
    const double* din[7];
    if (pinned) {   // gather the inputs in the pinned mirror: ONE copy to the device
        double* h = pinned;
        for (int i = 0; i < 7; ++i) {
            memcpy(h, in_host[i], in_counts[i] * sizeof(double));
            h += in_counts[i];
        }
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(staging, pinned, in_total * sizeof(double), hipMemcpyHostToDevice, st));
    }
    for (int i = 0; i < 7; ++i) {
        if (!pinned) TRAJOPT_HIP_CHECK(hipMemcpyAsync(d, in_host[i], in_counts[i] * sizeof(double), hipMemcpyHostToDevice, st));
        din[i] = d;
        d += in_counts[i];
    }
    // the staging keeps whatever the previous call left: every output (and the diverge word) starts at zero
    TRAJOPT_HIP_CHECK(hipMemsetAsync(staging + in_total, 0, (total - in_total) * sizeof(double), st));
    double* dout[10];
    for (int i = 0; i < 10; ++i) {
        dout[i] = d;
        d += out_counts[i];
    }
    double* d_lmbda = d;
    int32_t* d_div = reinterpret_cast<int32_t*>(d + 1);
    if (pinned) {   // (the memset above covered it; the value travels through the mirror like everything else)
        pinned[d_lmbda - staging] = lmbda;
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(d_lmbda, pinned + (d_lmbda - staging), sizeof(double), hipMemcpyHostToDevice, st));
    } else {
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(d_lmbda, &lmbda, sizeof(double), hipMemcpyHostToDevice, st));
    }
    trajopt_backward_io_t io{};
    io.Cxx = din[0]; io.cx = din[1]; io.Cuu = din[2]; io.cu = din[3]; io.Cxu = din[4]; io.A = din[5]; io.B = din[6];
    io.lmbda = d_lmbda;
    io.Qxx = dout[0]; io.Qux = dout[1]; io.Quu = dout[2]; io.qx = dout[3]; io.qu = dout[4]; io.V = dout[5];
    io.v = dout[6]; io.dV = dout[7]; io.K = dout[8]; io.kff = dout[9];
    io.diverge = d_div;
    const int rc = trajopt_ilqr_backward_pass_batched(&io, reg, n, m, T, 1, TRAJOPT_LAYOUT_PROBLEM_MAJOR, 0, nullptr, st);
    if (rc != TRAJOPT_OK) return rc;
    int32_t div = -1;
    if (pinned) {   // outputs | lmbda, diverge: ONE copy back, then scattered to the caller's arrays
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(pinned + in_total, staging + in_total, (total - in_total) * sizeof(double), hipMemcpyDeviceToHost, st));
        TRAJOPT_HIP_CHECK(hipStreamSynchronize(st));
        for (int i = 0; i < 10; ++i) memcpy(out_host[i], pinned + (dout[i] - staging), out_counts[i] * sizeof(double));
        memcpy(&div, pinned + (d_lmbda - staging) + 1, sizeof(int32_t));
    } else {
        for (int i = 0; i < 10; ++i)
            TRAJOPT_HIP_CHECK(hipMemcpyAsync(out_host[i], dout[i], out_counts[i] * sizeof(double), hipMemcpyDeviceToHost, st));
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(&div, d_div, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        TRAJOPT_HIP_CHECK(hipStreamSynchronize(st));
    }
    *diverge = div < 0 ? 0 : div;  // util.cpp:120,146
    if (failed) *failed = div >= 0 ? 1 : 0;
    return TRAJOPT_OK;
}

}  // namespace trajopt

extern "C" {

int trajopt_ilqr_backward_pass_host(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                    const double* Cxu, const double* A, const double* B, double lmbda, int reg,
                                    int n, int m, int T, double* Qxx, double* Qux, double* Quu, double* qx,
                                    double* qu, double* V, double* v, double* dV, double* K, double* kff,
                                    int* diverge, int* failed) {
    if (!Cxx || !cx || !Cuu || !cu || !Cxu || !A || !B || !Qxx || !Qux || !Quu || !qx || !qu || !V || !v || !dV ||
        !K || !kff || !diverge)
        return TRAJOPT_ERR_BAD_ARG;
    if (n <= 0 || m <= 0 || T <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    HostStage* h = nullptr;   // inputs | outputs | scalars
    const int rc = host_stage_locked(backward_host_staging_doubles(n, m, T), &h);
    if (rc != TRAJOPT_OK) return rc;
    return backward_host_on_staging(h->dev, h->pin, nullptr, Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, n, m, T, Qxx, Qux,
                                    Quu, qx, qu, V, v, dV, K, kff, diverge, failed);
}

// ---- allocation-free variant of trajopt_ilqr_backward_pass_host ----------------------------------------
}  // extern "C"

namespace trajopt {

// Device staging of the allocation-free entry point: a file-scope array of the code object, one copy per GPU the
// library runs on, held until the process ends.  In host code `scratch` is only a symbol: its per-device address
// comes from hipGetSymbolAddress (below) and everything - copies, memset, kernel arguments - uses that pointer.
constexpr size_t kStaticStagingDoubles = TRAJOPT_NOALLOC_MAX_DOUBLES;
static __device__ __align__(256) double scratch[kStaticStagingDoubles];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static double* g_scratch_ptr[kMaxDevices] = {};   // [device ordinal] -> that device's copy of `scratch`

// address of the current device's copy (resolved once per device, with that device current)
static int static_staging(double** out) {
    int dev = -1;
    TRAJOPT_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return TRAJOPT_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        double* ptr = nullptr;
        TRAJOPT_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
        if (!ptr) return TRAJOPT_ERR_HIP;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return TRAJOPT_OK;
}

}  // namespace trajopt

extern "C" {

int64_t trajopt_ilqr_backward_pass_host_noalloc_doubles(int n, int m, int T) {
    if (n <= 0 || m <= 0 || T <= 0) return TRAJOPT_ERR_BAD_ARG;
    return (int64_t)backward_host_staging_doubles(n, m, T);
}

int trajopt_ilqr_backward_pass_host_noalloc(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                            const double* Cxu, const double* A, const double* B, double lmbda,
                                            int reg, int n, int m, int T, double* Qxx, double* Qux, double* Quu,
                                            double* qx, double* qu, double* V, double* v, double* dV, double* K,
                                            double* kff, int* diverge, int* failed, void* stream) {
    if (!Cxx || !cx || !Cuu || !cu || !Cxu || !A || !B || !Qxx || !Qux || !Quu || !qx || !qu || !V || !v || !dV ||
        !K || !kff || !diverge)
        return TRAJOPT_ERR_BAD_ARG;
    if (n <= 0 || m <= 0 || T <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (backward_host_staging_doubles(n, m, T) > kStaticStagingDoubles) return TRAJOPT_ERR_TOO_LARGE;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    double* staging = nullptr;
    const int rc = static_staging(&staging);
    if (rc != TRAJOPT_OK) return rc;
    return backward_host_on_staging(staging, nullptr, as_stream(stream), Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, n, m, T,
                                    Qxx, Qux, Quu, qx, qu, V, v, dV, K, kff, diverge, failed);
}

int trajopt_riccati_backward_pass_host(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                       const double* Cxu, const double* A, const double* B, const double* c, int n,
                                       int m, int T, double* V, double* v, double* K, double* kff, int* status) {
    if (!Cxx || !cx || !Cuu || !cu || !Cxu || !A || !B || !c || !V || !v || !K || !kff) return TRAJOPT_ERR_BAD_ARG;
    if (n <= 0 || m <= 0 || T <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (trajopt_device_count() < 0) return TRAJOPT_ERR_NO_DEVICE;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const size_t in_counts[8] = {nn * (T + 1), (size_t)n * (T + 1), mm * (T + 1), (size_t)m * (T + 1), nm * (T + 1),
                                 nn * T, nm * T, (size_t)n * T};
    const double* in_host[8] = {Cxx, cx, Cuu, cu, Cxu, A, B, c};
    const size_t out_counts[4] = {nn * (T + 1), (size_t)n * (T + 1), nm * T, (size_t)m * T};
    double* out_host[4] = {V, v, K, kff};
    size_t total = 1;
    for (size_t x : in_counts) total += x;
    for (size_t x : out_counts) total += x;
    DevBuf buf;
    TRAJOPT_HIP_CHECK(buf.alloc(total));
    hipStream_t st = nullptr;
    double* d = buf.p;
    const double* din[8];
    for (int i = 0; i < 8; ++i) {
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(d, in_host[i], in_counts[i] * sizeof(double), hipMemcpyHostToDevice, st));
        din[i] = d;
        d += in_counts[i];
    }
    double* dout[4];
    for (int i = 0; i < 4; ++i) {
        dout[i] = d;
        d += out_counts[i];
    }
    int32_t* d_status = reinterpret_cast<int32_t*>(d);
    const int rc = trajopt_riccati_backward_pass_batched(din[0], din[1], din[2], din[3], din[4], din[5], din[6], din[7],
                                                         n, m, T, 1, TRAJOPT_LAYOUT_PROBLEM_MAJOR, dout[0], dout[1],
                                                         dout[2], dout[3], d_status, st);
    if (rc != TRAJOPT_OK) return rc;
    for (int i = 0; i < 4; ++i)
        TRAJOPT_HIP_CHECK(hipMemcpyAsync(out_host[i], dout[i], out_counts[i] * sizeof(double), hipMemcpyDeviceToHost, st));
    int32_t s = 0;
    TRAJOPT_HIP_CHECK(hipMemcpyAsync(&s, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TRAJOPT_HIP_CHECK(hipStreamSynchronize(st));
    if (status) *status = s;
    return TRAJOPT_OK;
}

}  // extern "C"
