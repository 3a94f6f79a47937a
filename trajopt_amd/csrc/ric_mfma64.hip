// Workgroup-per-problem time-varying LQR Riccati sweep on the fp64 matrix cores, for 16 < dx <= 64, du <= 16
// (BASELINE config C5: dx = 64, du = 16, T = 1000, 1024 problems).
//
// Row A9 of the hot path: /root/reference/trajopt/riccati/riccati.py:100-126 ("no-1/2, affine-c" convention),
// with V = Qxx + Qux' K as a matrix product.
//
// The state dimension is cut into 16 x 16 tiles of v_mfma_f64_16x16x4_f64; F = [A B] is (4 x 5) tiles.  A
// workgroup is four waves; wave w owns BLOCK COLUMN w of the value function:
//   Ph1  P(w, :) = V'(w, :) F          wave w needs row strip w of V', which (V' symmetric) is the D-layout image of
//                                      its own block column: V' never leaves the registers (riccati.py:108-110)
//   Ph2  Q(:, w) = C(:, w) + F' P(:, w) Qxx block column w, Qux tile w, and a quarter of Quu = Cuu + B'V'B
//   Ph3  Quu = L D L', K = -Quu^-1 Qux (lane = column), kff = -1/2 Quu^-1 qu                  (riccati.py:118-121)
//   Ph4  V(:, w) = Qxx(:, w) + K' Qux(w) (= (Qux' K)(:, w) by symmetry), v = qx + 2 Qux' kff   (riccati.py:123-124)
// F_t is staged in LDS once per step (B operands and, for F', A operands are read from there), P is exchanged
// through LDS, everything else stays in registers.  The loads of step t-1 are issued at the top of step t.
//
// What this kernel assumes beyond riccati.py:100-126 - PROBLEM_MAJOR layout; Cxx symmetric to round-off, as a Hessian
// is (the block-column trick reads V' as its own transpose); Quu positive definite (LDL' without pivoting, whereas
// np.linalg.inv at riccati.py:118 also inverts an indefinite matrix) - is CHECKED, per problem: a pre-pass compares
// every Cxx slice with its transpose (ric_symcheck_kernel) and a non-positive pivot stops the sweep; either way the
// problem is flagged and the generic kernel (bwd_wg.hip: pivoted LU, no symmetry assumption) then redoes exactly the
// flagged problems.  The results therefore do not depend on which kernel the dispatcher picks.
#include <mutex>
#include "smallmat.h"

namespace trajopt {

typedef double d4 __attribute__((ext_vector_type(4)));

struct Ric64Args {
    const double *Cxx, *cx, *Cuu, *cu, *Cxu, *A, *B, *c;
    double *V, *v, *K, *kff;
    int32_t* status;
    int32_t* redo;   // [nb_problems]: non-zero = this problem belongs to the generic kernel (set by the symmetry pre-pass or here)
    int n, m, T;
};

// Pre-pass: redo[p] = 1 when some slice of Cxx is not symmetric to round-off (|C_ij - C_ji| > 1e-13 max|C|).
// One workgroup per slice - the pass is one extra read of Cxx at HBM speed (C5: 33.6 GB, ~10 % of the sweep); the
// slice goes through LDS so that both triangles are read along rows.
struct SymArgs {
    const double* Cxx;
    int32_t* redo;
    int n, T;
};
__global__ __launch_bounds__(256) void ric_symcheck_kernel(SymArgs a) {
    __shared__ double tile[64 * 65];
    __shared__ double red[4];
    const int n = a.n, tid = threadIdx.x;
    const size_t p = blockIdx.x, nn = (size_t)n * n;
    const double* C = a.Cxx + (p * (size_t)(a.T + 1) + blockIdx.y) * nn;
    double mx = 0.0;
    for (int e = tid; e < (int)nn; e += 256) {
        const double x = C[e];
        tile[(e / n) * 65 + e % n] = x;
        mx = fmax(mx, fabs(x));
    }
    for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    const double tol = 1e-13 * fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    bool asym = false;
    for (int e = tid; e < (int)nn; e += 256) {
        const int i = e % n, j = e / n;
        const double d = fabs(tile[j * 65 + i] - tile[i * 65 + j]);
        asym = asym || !(d <= tol);   // NaN counts as asymmetric: the generic kernel decides what it means
    }
    if (asym) a.redo[p] = 1;
}

// Per-device flags of the problems handed to the generic kernel, in a static array of the code object (the batched
// entry points never allocate): resolved once per device like the staging of the allocation-free entry point.
constexpr int kRicMaxProblems = 1 << 18;
static __device__ int32_t ric_redo_flags[kRicMaxProblems];
static std::mutex g_ric_mutex;
static int32_t* g_ric_flags[64] = {};
static int ric_flags(int32_t** out) {
    int dev = -1;
    TRAJOPT_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return TRAJOPT_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_ric_mutex);
    if (!g_ric_flags[dev]) {
        int32_t* ptr = nullptr;
        TRAJOPT_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(ric_redo_flags)));
        if (!ptr) return TRAJOPT_ERR_HIP;
        g_ric_flags[dev] = ptr;
    }
    *out = g_ric_flags[dev];
    return TRAJOPT_OK;
}

constexpr int kLdk = 66;   // column stride (doubles) of the LDS images of F and P: conflict-free B-operand reads
constexpr int kLd16 = 17;  // column stride of the 16-row LDS images (Qux, K, Quu partials, L)

struct Ric64Lds {
    double F[81 * kLdk];       // [col][k]: cols 0..63 = A, 64..79 = B, 80 = c
    double P[64 * kLdk];       // [col][k]: P = V' A   (the B block column of P stays in registers)
    double Qux[64 * kLd16];    // Qux[i][j] at i + 17 j
    double K[64 * kLd16];      // K[i][j]   at i + 17 j
    double QuuP[4][16 * kLd16];  // per-wave partial sums of Quu, (row, col) at row + 17 col
    double L[4][16 * kLd16];   // per-wave copy of the unit lower factor, L[i][k] at i * 17 + k
    double w[64], qx[64], qu[16], kff[16];
};

__device__ __forceinline__ double bcast_lane(double x, int lane) {  // lane is wave-uniform
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double xor_sum16(double x) {  // sum over lanes c, c+16, c+32, c+48
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

__global__ __launch_bounds__(256) void ric_mfma64_kernel(Ric64Args a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Ric64Lds& S = *reinterpret_cast<Ric64Lds*>(smem_raw);
    const int n = a.n, m = a.m, T = a.T;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const size_t p = blockIdx.x;
    if (a.redo[p] != 0) return;   // an asymmetric Cxx: the generic kernel's problem (block-uniform)
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const double* gA = a.A + p * nn * T;
    const double* gB = a.B + p * nm * T;
    const double* gc = a.c + p * (size_t)n * T;
    const double* gCxx = a.Cxx + p * nn * (T + 1);
    const double* gCxu = a.Cxu + p * nm * (T + 1);
    const double* gCuu = a.Cuu + p * mm * (T + 1);
    const double* gcx = a.cx + p * (size_t)n * (T + 1);
    const double* gcu = a.cu + p * (size_t)m * (T + 1);
    double* oV = a.V + p * nn * (T + 1);
    double* ov = a.v + p * (size_t)n * (T + 1);
    double* oK = a.K + p * nm * T;
    double* okff = a.kff + p * (size_t)m * T;

    // ---- staging map of F_t = [A_t B_t c_t]: thread-private slots, fixed for the whole sweep -----------------
    // Slots beyond the matrix re-read element 0 and are parked in a pad word that nothing reads (k = 64 of col 0):
    // no per-slot predicates (20 of them as SGPR pairs made hipcc spill scalars through v_writelane).
    constexpr int kPark = 64;
    int srcA[16], offA[16], srcB[4], offB[4];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int idx = tid + 256 * s;
        const bool ok = idx < (int)nn;
        srcA[s] = ok ? idx : 0;
        offA[s] = ok ? (idx / n) * kLdk + idx % n : kPark;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        const bool ok = idx < (int)nm;
        srcB[s] = ok ? idx : 0;
        offB[s] = ok ? (64 + idx / n) * kLdk + idx % n : kPark;
    }
    const int srcc = tid < n ? tid : 0, offc = tid < n ? 80 * kLdk + tid : kPark;
    double fA[16], fB[4], fc;
    auto fetch_F = [&](int t) {  // global -> registers, unconditional
        const double* pa = gA + nn * (size_t)t;
        const double* pb = gB + nm * (size_t)t;
#pragma unroll
        for (int s = 0; s < 16; ++s) fA[s] = pa[srcA[s]];
#pragma unroll
        for (int s = 0; s < 4; ++s) fB[s] = pb[srcB[s]];
        fc = gc[(size_t)n * t + srcc];
    };
    auto stage_F = [&]() {       // registers -> LDS
#pragma unroll
        for (int s = 0; s < 16; ++s) S.F[offA[s]] = fA[s];
#pragma unroll
        for (int s = 0; s < 4; ++s) S.F[offB[s]] = fB[s];
        S.F[offc] = fc;
    };

    // zero the padding of the F image once (valid entries are overwritten every step), P needs none
    for (int i = tid; i < 81 * kLdk; i += 256) S.F[i] = 0.0;
    __syncthreads();
    fetch_F(T - 1);
    stage_F();

    // ---- terminal condition (riccati.py:104-105): block column w of V, own segment of v ----------------------
    const int col = 16 * w + c;          // the state index this lane owns as a column
    const bool colok = col < n;
    d4 Vt[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * kt + g + 4 * r;
            double x = 0.0;
            if (colok && row < n) {
                x = gCxx[nn * (size_t)T + col + (size_t)n * row];  // (symmetric) read along the contiguous index
                oV[nn * (size_t)T + col + (size_t)n * row] = x;
            }
            Vt[kt][r] = x;
        }
    double vown = colok ? gcx[(size_t)n * T + col] : 0.0;
    if (g == 0 && colok) ov[(size_t)n * T + col] = vown;
    __syncthreads();

    int bad = -1;
    for (int t = T - 1; t >= 0; --t) {
        // ================= Ph1: prefetch, w = 2 V'c + v', P = V' F ==========================================
        fetch_F(t >= 1 ? t - 1 : 0);
        d4 Q[5];
#pragma unroll
        for (int I = 0; I < 4; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + g + 4 * r;
                const bool ok = colok && row < n;
                const double x = gCxx[nn * (size_t)t + (ok ? col + (size_t)n * row : 0)];
                Q[I][r] = ok ? x : 0.0;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // Qux tile starts from Cxu'  (riccati.py:110): element (i, col) = Cxu[col][i]
            const int i = g + 4 * r;
            const bool ok = colok && i < m;
            const double x = gCxu[nm * (size_t)t + (ok ? col + (size_t)n * i : 0)];
            Q[4][r] = ok ? x : 0.0;
        }
        d4 qp = {0.0, 0.0, 0.0, 0.0};  // this wave's part of Quu; wave 0 starts from Cuu
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = g + 4 * r;
            const bool ok = w == 0 && i < m && c < m;
            const double x = gCuu[mm * (size_t)t + (ok ? i + (size_t)m * c : 0)];
            qp[r] = ok ? x : 0.0;
        }
        double cq = 0.0;  // cx / cu element for the q vectors (threads 0..159, two per column)
        {
            const int qc = tid >> 1;
            if (tid < 160) {
                const bool isx = qc < 64;
                const bool ok = isx ? qc < n : (qc - 64) < m;
                const double x = isx ? gcx[(size_t)n * t + (ok ? qc : 0)] : gcu[(size_t)m * t + (ok ? qc - 64 : 0)];
                cq = ok ? x : 0.0;
            }
        }
        {
            double vc = 0.0;  // (V'c)[col] = sum_k V'[col][k] c[k], k = 16 kt + 4 s + g on this lane
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 4; ++s) vc = fma(Vt[kt][s], S.F[80 * kLdk + 16 * kt + 4 * s + g], vc);
            vc = xor_sum16(vc);
            if (g == 0) S.w[col] = fma(2.0, vc, vown);  // riccati.py:112-116
        }
        d4 Pw4;
#pragma unroll
        for (int J = 0; J < 5; ++J) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[kt][s], S.F[(16 * J + c) * kLdk + 16 * kt + 4 * s + g], acc,
                                                               0, 0, 0);
            if (J < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) S.P[(16 * J + c) * kLdk + 16 * w + g + 4 * r] = acc[r];
            } else {
                Pw4 = acc;
            }
        }
        __syncthreads();

        // ================= Ph2: Q(:, w) = C(:, w) + F' P(:, w), Quu part, q vectors ===========================
        {
            d4 Pc[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 4; ++s) Pc[kt][s] = S.P[(16 * w + c) * kLdk + 16 * kt + 4 * s + g];
#pragma unroll
            for (int I = 0; I < 5; ++I)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        Q[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.F[(16 * I + c) * kLdk + 16 * kt + 4 * s + g], Pc[kt][s],
                                                                    Q[I], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qp = __builtin_amdgcn_mfma_f64_16x16x4f64(S.F[(64 + c) * kLdk + 16 * w + 4 * s + g], Pw4[s], qp, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            S.QuuP[w][(g + 4 * r) + kLd16 * c] = qp[r];
            S.Qux[(g + 4 * r) + kLd16 * col] = Q[4][r];
        }
        if (tid < 160) {  // [qx; qu] = [cx; cu] + F' w   (riccati.py:112-116)
            const int qc = tid >> 1, half = tid & 1;
            const double* fcol = &S.F[qc * kLdk + 32 * half];
            const double* wv = &S.w[32 * half];
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < 32; k += 2) {
                s0 = fma(fcol[k], wv[k], s0);
                s1 = fma(fcol[k + 1], wv[k + 1], s1);
            }
            double sum = s0 + s1;
            sum += __shfl_xor(sum, 1, 64);
            if (half == 0) {
                if (qc < 64) S.qx[qc] = cq + sum;
                else S.qu[qc - 64] = cq + sum;
            }
        }
        __syncthreads();

        // ================= Ph3: Quu = L D L' (every wave, redundantly), K (wave 0), kff (wave 1) ==============
        stage_F();  // the F image of this step is dead: bring in step t - 1
        double row[16], invd[16];
        bool ok = true;
        {
            const int r = c;  // lane holds row r of Quu (replicated over g); rows/cols >= m are identity padding
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double x = 0.0;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) x += S.QuuP[ww][r + kLd16 * j];
                row[j] = (r < m && j < m) ? x : (r == j ? 1.0 : 0.0);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double dk = bcast_lane(row[k], k);
                ok = ok && (dk > 0.0);
                const double ik = rcp_nr(dk);
                invd[k] = ik;
                const double l = (r > k) ? row[k] * ik : 0.0;
#pragma unroll
                for (int j = k + 1; j < 16; ++j) row[j] = fma(-l, bcast_lane(row[j], k), row[j]);
                if (r > k) row[k] = l;
            }
            if (g == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) S.L[w][r * kLd16 + k] = row[k];
            }
        }
        if (!ok) {  // identical in every wave: block-uniform exit
            bad = t;
            break;
        }
        if (w < 2) {
            double y[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) y[i] = (w == 0) ? S.Qux[i + kLd16 * lane] : S.qu[i];
            const double* Lw = S.L[w];
#pragma unroll
            for (int i = 1; i < 16; ++i)
#pragma unroll
                for (int k = 0; k < i; ++k) y[i] = fma(-Lw[i * kLd16 + k], y[k], y[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) y[i] *= invd[i];
#pragma unroll
            for (int i = 14; i >= 0; --i)
#pragma unroll
                for (int k = i + 1; k < 16; ++k) y[i] = fma(-Lw[k * kLd16 + i], y[k], y[i]);
            if (w == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) S.K[i + kLd16 * lane] = -y[i];  // riccati.py:120
            } else if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) S.kff[i] = -0.5 * y[i];         // riccati.py:121
            }
        }
        __syncthreads();

        // ================= Ph4: V(:, w) = Qxx(:, w) + K' Qux(w), v = qx + 2 Qux' kff, outputs ================
#pragma unroll
        for (int I = 0; I < 4; ++I)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                Q[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.K[(4 * s + g) + kLd16 * (16 * I + c)], Q[4][s], Q[I], 0, 0, 0);
        {
            double part = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) part = fma(Q[4][r], S.kff[g + 4 * r], part);
            part = xor_sum16(part);
            vown = colok ? fma(2.0, part, S.qx[col]) : 0.0;  // riccati.py:124
        }
#pragma unroll
        for (int I = 0; I < 4; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rw = 16 * I + g + 4 * r;
                const bool okv = colok && rw < n;
                const double x = okv ? Q[I][r] : 0.0;  // padding stays exactly zero
                Vt[I][r] = x;
                if (okv) oV[nn * (size_t)t + col + (size_t)n * rw] = x;  // V' = V: stored along the contiguous index
            }
        if (g == 0 && colok) ov[(size_t)n * t + col] = vown;
        for (int idx = tid; idx < (int)nm; idx += 256) {
            const int i = idx % m, j = idx / m;
            oK[nm * (size_t)t + idx] = S.K[i + kLd16 * j];
        }
        if (tid < m) okff[(size_t)m * t + tid] = S.kff[tid];
    }
    // a non-positive pivot: Quu is singular or indefinite - the generic kernel's pivoted LU tells which (riccati.py:118)
    if (tid == 0) {
        if (bad >= 0) a.redo[p] = 1;
        if (a.status) a.status[p] = 0;
    }
}

// handled = false when the shape / layout is outside this kernel's range.
int launch_riccati_mfma64(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                          const double* A, const double* B, const double* c, int n, int m, int T, int64_t NP,
                          int layout, double* V, double* v, double* K, double* kff, int32_t* status, hipStream_t st,
                          bool* handled) {
    *handled = false;
    if (n <= 16 || n > 64 || m < 1 || m > 16) return TRAJOPT_OK;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && NP != 1) return TRAJOPT_OK;
    if (NP > 0x7fffffff) return TRAJOPT_OK;
    if (NP > kRicMaxProblems) return TRAJOPT_OK;   // (more problems than flags: the generic kernel takes them all)
    int32_t* redo = nullptr;
    const int rc = ric_flags(&redo);
    if (rc != TRAJOPT_OK) return rc;
    *handled = true;
    TRAJOPT_HIP_CHECK(hipMemsetAsync(redo, 0, (size_t)NP * sizeof(int32_t), st));
    if (T + 1 > 65535) { *handled = false; return TRAJOPT_OK; }   // (grid rows of the pre-pass)
    const SymArgs sa{Cxx, redo, n, T};
    hipLaunchKernelGGL(ric_symcheck_kernel, dim3((unsigned)NP, (unsigned)(T + 1)), dim3(256), 0, st, sa);
    Ric64Args a{Cxx, cx, Cuu, cu, Cxu, A, B, c, V, v, K, kff, status, redo, n, m, T};
    const size_t lds = sizeof(Ric64Lds);
    TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)ric_mfma64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
    hipLaunchKernelGGL(ric_mfma64_kernel, dim3((unsigned)NP), dim3(256), lds, st, a);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    // the flagged problems, by the kernel without those assumptions (a workgroup of an unflagged problem leaves at once)
    return launch_riccati_wg(Cxx, cx, Cuu, cu, Cxu, A, B, c, n, m, T, NP, TRAJOPT_LAYOUT_PROBLEM_MAJOR, V, v, K, kff, status, st, redo);
}

}  // namespace trajopt
