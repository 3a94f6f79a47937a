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
//   Ph3  block elimination of [Quu | Qux(:, w) | qu] with 4 x 4 pivot blocks, every product a matrix-core instruction:
//        Y = L^-1 Qux, Z = D^-1 Y (forward), K = -U^-1 Z, kff = -1/2 U^-1 zq (backward)        (riccati.py:118-121)
//   Ph4  V(:, w) = Qxx(:, w) - Z' Y(:, w) (= Qxx + Qux' K), v = qx - Y' zq (= qx + 2 Qux' kff)  (riccati.py:123-124)
// F_t is staged in LDS once per step (B operands and, for F', A operands are read from there), P is exchanged
// through LDS, everything else stays in registers.  The loads of step t-1 are issued at the top of step t.
//
// What this kernel assumes beyond riccati.py:100-126 - PROBLEM_MAJOR layout; Cxx symmetric to round-off, as a Hessian
// is (the block-column trick reads V' as its own transpose); Quu positive definite (LDL' without pivoting, whereas
// np.linalg.inv at riccati.py:118 also inverts an indefinite matrix) - is CHECKED, per problem and on the fly: every
// slice of Cxx is read both ways and compared (|C_ij - C_ji| <= 1e-13 max |C|: 16 more loads a lane and step, no extra
// pass over the 33.6 GB of C5's Cxx) and a non-positive pivot stops the sweep; either way the problem is flagged and
// the generic kernel (bwd_wg.hip: pivoted LU, no symmetry assumption) then redoes exactly the flagged problems.  The
// results therefore do not depend on which kernel the dispatcher picks.
#include <mutex>
#include "smallmat.h"

namespace trajopt {

typedef double d4 __attribute__((ext_vector_type(4)));

struct Ric64Args {
    const double *Cxx, *cx, *Cuu, *cu, *Cxu, *A, *B, *c;
    double *V, *v, *K, *kff;
    int32_t* status;
    // [nb_problems], only when status is NULL: non-zero = this problem belongs to the generic kernel (set by the sweep:
    // asymmetric Cxx slice or non-positive pivot).  With a status array the flag travels in status[p] (kRicRedo), memory
    // the CALLER owns: calls that overlap on one GPU then share nothing.
    int32_t* redo;
    int n, m, T;
};

// status[p] of a problem the sweep hands to the generic kernel; that kernel (bwd_wg.hip, `only`) runs the problems whose
// entry is non-zero and overwrites it with the final status.
constexpr int32_t kRicRedo = INT32_MIN;
// Callers that pass status = NULL have no memory of their own for that flag: for them it lives in a per-device static
// array of the code object (the batched entry points never allocate), resolved once per device like the staging of the
// allocation-free entry point - and calls WITHOUT a status array that overlap on one GPU race on it (trajopt_amd.h).
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

#ifndef RIC_SPREAD
#define RIC_SPREAD 1   // 1: the phase's global loads are issued between its first matrix-core instructions (below)
#endif
#ifndef RIC_DIAG
#define RIC_DIAG 0   // timing experiments only: 1 = per-phase clock ticks of wave 0 written over v[p][0..7][0] (wrong results)
#endif
#if RIC_DIAG
#define RIC_TICK(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tick_[k] += now_ - last_; last_ = now_; }
#else
#define RIC_TICK(k)
#endif
constexpr int kLdk = 66;   // column stride (doubles) of the LDS images of F and P: conflict-free 16-byte operand reads
// Position of inner index k inside a column of the F / P images (and of S.w): the matrix-core operand of lane (c, g) for
// k-step (kt, s) is element k = 16 kt + 4 s + g, so a lane's 16 operands of a tile row are k = g, g + 4, ... - stored
// next to each other (16 g + k / 4) they are eight 16-byte LDS reads issued ahead of the 16 MFMAs they feed, where one
// 8-byte read per MFMA, each waited for on the spot, left the matrix pipe idle half of the time.
__host__ __device__ constexpr int pk(int k) { return (k & 3) * 16 + (k >> 2); }
typedef double r64_d2 __attribute__((ext_vector_type(2)));
// the 16 operands of lane (c, g) from column `col` of an image
__device__ __forceinline__ void ld16(double (&f)[16], const double* img, int col, int g) {
    const r64_d2* src = reinterpret_cast<const r64_d2*>(img + col * kLdk + 16 * g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const r64_d2 x = src[e];
        f[2 * e] = x[0];
        f[2 * e + 1] = x[1];
    }
}
constexpr int kLd16 = 17;  // column stride of the 16-row LDS images (Qux, K, Quu partials, L)

struct Ric64Lds {
    double F[81 * kLdk];       // [col][k]: cols 0..63 = A, 64..79 = B, 80 = c
    double P[64 * kLdk];       // [col][k]: P = V' A   (the B block column of P stays in registers)
    double Ko[64 * kLd16];     // the gains K[i][j] at i + 17 j, for the output stores
    double K[64 * kLd16];      // -Z[i][j] = -(D^-1 L^-1 Qux)[i][j] at i + 17 j: the A operand of the V update
    double QuuP[4][16 * kLd16];  // per-wave partial sums of Quu, (row, col) at row + 17 col
    double U[4][16 * kLd16];   // per-wave copy of the unit upper block factor, U[i][j] at i + 17 j
    double piv[4][16];         // per wave: the 4 x 4 pivot block of the current elimination step, (a, b) at a + 4 b
    double w[64], qx[64], qu[16], kff[16], zq[16];
    double X[64 * kLdk];       // [col][row]: Cxx_t as it stands in memory, element (row, col) at pk(row) of column col
    double cmax[4];            // per-wave max |Cxx_t| (symmetry check)
    int asym;                  // set when a slice of Cxx fails the symmetry check
};

typedef unsigned r64_v2u __attribute__((ext_vector_type(2)));
constexpr unsigned kR64NoStore = 0xfffffff8u;  // beyond any per-problem slice (the launcher checks their sizes)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t r64_rsrc(const double* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double r64_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {   // 0.0 beyond the range (voff)
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void r64_st(double x, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(r64_v2u, x), r, off, 0, 0);
}

__device__ __forceinline__ void r64_st_s(double x, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {   // range check on voff alone
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(r64_v2u, x), r, voff, soff, 0);
}

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

// LDS round trip inside ONE wave (its LDS accesses execute in program order): only the compiler must keep the order
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// FULL: dx = 64, du = 16 (C5) known at compile time - every padding predicate, lane mask and offset product of the time
// loop folds away (as run-time values their masks lived in ~40 scalar registers, spilled to lanes of a vector register and
// read back inside the matrix-core slots); the other instantiation takes any 16 < dx <= 64, du <= 16.
template <bool FULL>
__global__ __launch_bounds__(256) void ric_mfma64_kernel(Ric64Args a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Ric64Lds& S = *reinterpret_cast<Ric64Lds*>(smem_raw);
    const int n = FULL ? 64 : a.n, m = FULL ? 16 : a.m, T = a.T;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const size_t p = blockIdx.x;
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
    // Outputs go through range-checked descriptors of this problem's slices: a thread with nothing to store passes an
    // offset beyond the range and the write is dropped, so every store of the time loop is an unconditional instruction
    // (stores under branches cannot be counted by hipcc's wait-count pass, which then waits for all of them).
    const __amdgpu_buffer_rsrc_t rV = r64_rsrc(oV, (unsigned)(nn * (size_t)(T + 1) * 8));
    const __amdgpu_buffer_rsrc_t rv_ = r64_rsrc(ov, (unsigned)((size_t)n * (T + 1) * 8));
    const __amdgpu_buffer_rsrc_t rK = r64_rsrc(oK, (unsigned)(nm * (size_t)T * 8));
    const __amdgpu_buffer_rsrc_t rkff = r64_rsrc(okff, (unsigned)((size_t)m * T * 8));
    // The cost inputs likewise: a padding lane passes the out-of-range offset and the hardware returns 0.0 - no select
    // on the loaded value (hipcc scheduled those selects right behind the loads, i.e. waited out the HBM latency at the
    // top of every step instead of where the values are first used, a phase later).
    const __amdgpu_buffer_rsrc_t rCxx = r64_rsrc(gCxx, (unsigned)(nn * (size_t)(T + 1) * 8));
    const __amdgpu_buffer_rsrc_t rCxu = r64_rsrc(gCxu, (unsigned)(nm * (size_t)(T + 1) * 8));
    const __amdgpu_buffer_rsrc_t rCuu = r64_rsrc(gCuu, (unsigned)(mm * (size_t)(T + 1) * 8));
    const int wu = __builtin_amdgcn_readfirstlane(tid) >> 6;   // the wave, as a scalar
    const __amdgpu_buffer_rsrc_t rcq = r64_rsrc(wu < 2 ? gcx : gcu, wu < 3 ? (unsigned)((size_t)(wu < 2 ? n : m) * (T + 1) * 8) : 0u);

    // ---- staging map of F_t = [A_t B_t c_t]: thread-private slots, fixed for the whole sweep -----------------
    // Slots beyond the matrix re-read element 0 and are parked in a pad word that nothing reads (k = 64 of col 0):
    // no per-slot predicates (20 of them as SGPR pairs made hipcc spill scalars through v_writelane).
    constexpr int kPark = 64;
    int offA[16], offB[4];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int idx = tid + 256 * s;
        offA[s] = (FULL || idx < (int)nn) ? (idx / n) * kLdk + pk(idx % n) : kPark;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        offB[s] = (FULL || idx < (int)nm) ? (64 + idx / n) * kLdk + pk(idx % n) : kPark;
    }
    const int offc = tid < n ? 80 * kLdk + pk(tid) : kPark;
    // Every load of the time loop is  descriptor of the problem's array + loop-invariant lane offset (VGPR) + scalar
    // offset of the step: no vector address arithmetic inside the loop.  A parked slot reads whatever lies at its
    // offset (the next slice; zero beyond the array).
    const __amdgpu_buffer_rsrc_t rA = r64_rsrc(gA, (unsigned)(nn * (size_t)T * 8));
    const __amdgpu_buffer_rsrc_t rB = r64_rsrc(gB, (unsigned)(nm * (size_t)T * 8));
    const __amdgpu_buffer_rsrc_t rc_ = r64_rsrc(gc, (unsigned)((size_t)n * T * 8));
    const unsigned vo_tid = (unsigned)tid * 8u, vo_c = (unsigned)(tid < n ? tid : 0) * 8u;
    double fA[16], fB[4], fc;
    auto fetch_F = [&](int t) {  // global -> registers, unconditional
        const unsigned tu = (unsigned)__builtin_amdgcn_readfirstlane(t);
#pragma unroll
        for (int s = 0; s < 16; ++s) fA[s] = r64_ld(rA, vo_tid, (tu * (unsigned)nn + 256u * s) * 8u);
#pragma unroll
        for (int s = 0; s < 4; ++s) fB[s] = r64_ld(rB, vo_tid, (tu * (unsigned)nm + 256u * s) * 8u);
        fc = r64_ld(rc_, vo_c, tu * (unsigned)n * 8u);
    };
    // one piece of the same prefetch: loads i of 30 (fetch_F = pieces 0 .. 20)
    auto fetch_F_piece = [&](unsigned tu, int i) {
        if (i < 16) fA[i] = r64_ld(rA, vo_tid, (tu * (unsigned)nn + 256u * i) * 8u);
        else if (i < 20) fB[i - 16] = r64_ld(rB, vo_tid, (tu * (unsigned)nm + 256u * (i - 16)) * 8u);
        else if (i == 20) fc = r64_ld(rc_, vo_c, tu * (unsigned)n * 8u);
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
    for (int i = tid; i < 64 * kLdk; i += 256) S.X[i] = 0.0;
    __syncthreads();
    fetch_F(T - 1);
    stage_F();

    // ---- terminal condition (riccati.py:104-105): block column w of V, own segment of v ----------------------
    const int col = 16 * w + c;          // the state index this lane owns as a column
    const bool colok = FULL || col < n;   // (FULL: a block of 256 threads covers exactly the 64 columns)
    // lane offsets of the cost inputs: element (row, col) of this lane's block column read along the contiguous index,
    // i.e. as C[col][row] (voT) - and, for the symmetry check, as it stands (voS); padding lanes pass the dropped offset
    unsigned voT[16], voS[16], voXu[4], voUu[4];
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + g + 4 * r;
            const bool ok = colok && (FULL || row < n);
            voT[4 * I + r] = ok ? (unsigned)(col + n * row) * 8u : kR64NoStore;
            voS[4 * I + r] = ok ? (unsigned)(row + n * col) * 8u : kR64NoStore;
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = g + 4 * r;
        voXu[r] = (colok && (FULL || i < m)) ? (unsigned)(col + n * i) * 8u : kR64NoStore;
        voUu[r] = (w == 0 && i < m && c < m) ? (unsigned)(i + m * c) * 8u : kR64NoStore;
    }
    const unsigned voRow = (FULL || lane < n) ? (unsigned)lane * 8u : kR64NoStore;
    unsigned vocq;
    {
        const int qc = tid >> 1, el = qc < 64 ? qc : qc - 64, ne = qc < 64 ? n : m;
        vocq = (tid < 160 && el < ne) ? (unsigned)el * 8u : kR64NoStore;
    }
    // Symmetry of a slice of Cxx (the block-column trick reads V' as its own transpose): |C_ij - C_ji| <= 1e-13 max |C|,
    // checked on the fly - each lane compares the transposed read it uses with the straight read of the same element.
    // sym_lane: this lane's largest difference (NaN / Inf poison it) and its largest magnitude.
    auto sym_lane = [&](const d4 (&Qt)[5], const double (&xs)[16], double& dmax, double& cmax) {
        double dm = 0.0, cm = 0.0, poison = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double d = Qt[e >> 2][e & 3] - xs[e];
            dm = fmax(dm, fabs(d));
            cm = fmax(cm, fabs(xs[e]));
            poison = fma(d, 0.0, poison);
        }
        dmax = dm + poison;
        for (int o = 32; o >= 1; o >>= 1) cm = fmax(cm, __shfl_xor(cm, o, 64));
        cmax = cm;
    };
    if (tid == 0) S.asym = 0;
    d4 Vt[4];
    {
        d4 Qt[5];
        double xs[16], dmax, cmax;
        const unsigned sT = (unsigned)(nn * (size_t)T * 8);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            Qt[e >> 2][e & 3] = r64_ld(rCxx, voT[e], sT);
            xs[e] = r64_ld(rCxx, voS[e], sT);
        }
        sym_lane(Qt, xs, dmax, cmax);
        if (lane == 0) S.cmax[w] = cmax;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            Vt[e >> 2][e & 3] = Qt[e >> 2][e & 3];
            r64_st(Qt[e >> 2][e & 3], rV, voT[e] != kR64NoStore ? voT[e] + sT : kR64NoStore);
        }
        __syncthreads();
        const double tol = 1e-13 * fmax(fmax(S.cmax[0], S.cmax[1]), fmax(S.cmax[2], S.cmax[3]));
        if (!(dmax <= tol)) S.asym = 1;
    }
    double vown = colok ? gcx[(size_t)n * T + col] : 0.0;
    if (g == 0 && colok) ov[(size_t)n * T + col] = vown;
    __syncthreads();
    if (S.asym != 0) {   // block-uniform: the generic kernel takes this problem
        if (tid == 0) {
            if (a.status) a.status[p] = kRicRedo;
            else a.redo[p] = 1;
        }
        return;
    }

    // 0 / 1 masks that place the ten distinct entries of a symmetric 4 x 4 matrix Pi into the matrix-core A operand
    // A[row c][k-index g] = Pi(c, g), c < 4 (Ph3): entry (a, b), a >= b, goes to the lanes (c, g) = (a, b) and (b, a)
    double mk[10];
    {
        int e = 0;
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
            for (int a2 = b2; a2 < 4; ++a2) mk[e++] = ((c == a2 && g == b2) || (c == b2 && g == a2)) ? 1.0 : 0.0;
    }
    int bad = -1;
    // nothing of the set-up is in flight when the loop starts: its wait counts then cover the loop's own operations only
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#if RIC_DIAG
    unsigned long long tick_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
    // Cxx_t: wave w fetches columns 16 w .. 16 w + 15, lane = row - whole 512-byte rows, every element once - and hands
    // them to the block through LDS at the end of Ph1; Ph2 takes its tile AND the mirrored elements (the symmetry check)
    // from there.  The slice of step t - 1 is requested in Ph3 of step t, two phases before it is staged: requested at the
    // top of Ph1 (behind the 21 loads of F, which retire in order before it) the stage waited ~2k cycles a step for it.
    double xs[16];
    auto fetch_X = [&](int t) {
        const unsigned tu = (unsigned)__builtin_amdgcn_readfirstlane(t);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
            xs[jj] = r64_ld(rCxx, (FULL || (16 * wu + jj) < n) ? voRow : kR64NoStore, (tu * (unsigned)nn + (unsigned)(n * (16 * wu + jj))) * 8u);
    };
    fetch_X(T - 1);
    // The outputs of a step are stored one piece a slot from the shadows of the NEXT step's Ph1 (Vt, vown and the LDS
    // images of K, kff stand unchanged until that step's Ph3 / Ph4) - behind the V update they cost Ph4 ~1.5k cycles a step
    // with nothing to overlap.  Piece j < 16: element (16 I + g + 4 r, col) of V, j = 4 I + r, stored along the contiguous
    // index (V' = V); 16..19: four gains a thread; 20: v; 21: kff.  `have` = there is such a step (not in the first
    // iteration, whose "previous step" is the terminal slice stored above).
    unsigned okbits = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) okbits |= (colok && (FULL || (16 * (j >> 2) + g + 4 * (j & 3)) < n)) ? (1u << j) : 0u;
    const unsigned voV0 = (unsigned)(col + n * g) * 8u;
    constexpr int kOutPieces = 22;
    auto out_piece = [&](int j, unsigned tt, bool have) {
        if (j < 16) {
            const bool okv = have && ((okbits >> j) & 1u);
            r64_st_s(Vt[j >> 2][j & 3], rV, okv ? voV0 : kR64NoStore, (tt * (unsigned)nn + (unsigned)(n * (16 * (j >> 2) + 4 * (j & 3)))) * 8u);
        } else if (j < 20) {   // dx du <= 1024 gains: four per thread
            const int idx = tid + 256 * (j - 16);
            const bool okk = FULL || idx < (int)nm;
            const int i = okk ? idx % m : 0, jj = okk ? idx / m : 0;
            r64_st_s(S.Ko[i + kLd16 * jj], rK, (okk && have) ? (unsigned)idx * 8u : kR64NoStore, tt * (unsigned)nm * 8u);
        } else if (j == 20) {
            r64_st_s(vown, rv_, (g == 0 && colok && have) ? (unsigned)col * 8u : kR64NoStore, tt * (unsigned)n * 8u);
        } else {
            r64_st_s(S.kff[tid < m ? tid : 0], rkff, (tid < m && have) ? (unsigned)tid * 8u : kR64NoStore, tt * (unsigned)m * 8u);
        }
    };
    // Block back substitution W = U^-1 Z (U unit upper, this wave's copy in S.U): rows of block j are final once the blocks
    // right of it have been applied; A[row c][k-index g] = U(c, 4 j + g) for the rows c above block j.  K = -W,
    // kff = -1/2 U^-1 zq (riccati.py:120-121).  Only the OUTPUTS need them, so the three steps (two dependent MFMAs each)
    // and the hand-over to LDS are spread over the first slots of the next step's Ph1; the stores follow in its Ph2.
    d4 Zt = {0.0, 0.0, 0.0, 0.0}, Zq4 = {0.0, 0.0, 0.0, 0.0};
    auto back_sub = [&](int j) {
        const double u = S.U[w][c + kLd16 * (4 * j + g)];
        const double nU = (c < 4 * j) ? -u : 0.0;
        Zt = __builtin_amdgcn_mfma_f64_16x16x4f64(nU, Zt[j], Zt, 0, 0, 0);
        Zq4 = __builtin_amdgcn_mfma_f64_16x16x4f64(nU, Zq4[j], Zq4, 0, 0, 0);
    };
    auto gains_to_lds = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) S.Ko[(g + 4 * r) + kLd16 * col] = -Zt[r];
        if (w == 1 && c == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S.kff[g + 4 * r] = -0.5 * Zq4[r];
        }
    };
    for (int t = T - 1; t >= 0; --t) {
        // ================= Ph1: prefetch, w = 2 V'c + v', P = V' F ==========================================
        // This phase's 30 global loads (F_{t-1}: 21; the Qux tile's start Cxu', riccati.py:110: element (i, col) =
        // Cxu[col][i]; this wave's part of Quu - wave 0 starts from Cuu, the others read beyond the range: 0.0; the cx /
        // cu element for the q vectors - threads 0..159, two per column; waves 0, 1: cx, wave 2: cu) are not needed before
        // Ph2 / Ph3.  The wave issues in order: what stands between two matrix-core instructions runs while the first one
        // occupies the pipe (64 cycles), what stands in front of the whole run of them does not - so the loads and their
        // scalar address arithmetic are issued one at a time between the first 64 MFMAs (piece() below, every other slot)
        // instead of as a 230-instruction prologue.
        const unsigned tu = (unsigned)__builtin_amdgcn_readfirstlane(t);
        const unsigned tp = (unsigned)__builtin_amdgcn_readfirstlane(t >= 1 ? t - 1 : 0);
        const unsigned tn = (unsigned)__builtin_amdgcn_readfirstlane(t + 1);   // the step whose outputs go out under this Ph1
        const bool have_out = t < T - 1;
        d4 Q[5];
        d4 qp;
        double cq;
        auto piece = [&](int i) {
            if (i < 21) fetch_F_piece(tp, i);
            else if (i < 25) Q[4][i - 21] = r64_ld(rCxu, voXu[i - 21], tu * (unsigned)nm * 8u);
            else if (i < 29) qp[i - 25] = r64_ld(rCuu, voUu[i - 25], tu * (unsigned)mm * 8u);
            else if (i == 29) cq = r64_ld(rcq, vocq, tu * (unsigned)(wu < 2 ? n : m) * 8u);
        };
        if (!RIC_SPREAD) {
#pragma unroll
            for (int i = 0; i < 30; ++i) piece(i);
#pragma unroll
            for (int j = 0; j < 16; ++j) out_piece(j, tn, have_out);
            out_piece(20, tn, have_out);
            back_sub(3);
            back_sub(2);
            back_sub(1);
            gains_to_lds();
        }
        // The B operands (a lane's 16 values of a column of the F image: eight 16-byte LDS reads) are streamed, each pair
        // read a few slots ahead of the MFMAs that use it (see Ph2).
        auto prF = [&](int colx, int i) {   // pair i (elements 2 i, 2 i + 1) of this lane's operands of column colx
            return *reinterpret_cast<const r64_d2*>(&S.F[colx * kLdk + 16 * g + 2 * i]);
        };
        r64_d2 T0[8], T1[8], T2[8], T3[8], TB[8], CC[8];
        T0[0] = prF(c, 0);
        T1[0] = prF(16 + c, 0);
        CC[0] = prF(80, 0);         // the c_t column (every lane the same column): for w below
        // Two tiles at a time, their MFMAs alternating: a chain of 16 MFMAs on one accumulator leaves the matrix pipe
        // idle between dependent issues, two independent chains keep it busy.
        d4 Pw4, pa, pb;   // (pa, pb: finished tiles on their way to LDS)
        auto put_P = [&](const d4& acc, int J) {   // rows 16 w + g + 4 r of column 16 J + c: positions 16 g + 4 w + r
            r64_d2* dst = reinterpret_cast<r64_d2*>(&S.P[(16 * J + c) * kLdk + 16 * g + 4 * w]);
            dst[0] = r64_d2{acc[0], acc[1]};
            dst[1] = r64_d2{acc[2], acc[3]};
        };
        {
            // Everything else the phase has to do rides in the shadow of these 32 MFMAs, one piece a slot (RIC_SPREAD): the
            // global loads, w = 2 V'c + v' (riccati.py:112-116; a chain of 16 FMAs that each wait for an LDS operand - in
            // front of the MFMAs it cost the phase ~600 cycles a step) and the operand reads of the next two tiles.
            double vc = 0.0;  // (V'c)[col] = sum_k V'[col][k] c[k], k = 16 kt + 4 s + g on this lane
            auto side = [&](int slot) {
                if (slot >= 2 && slot < 18) {
                    const int el = slot - 2;
                    vc = fma(Vt[el >> 2][el & 3], CC[el >> 1][el & 1], vc);
                    asm volatile("" : "+v"(vc));
                }
                if (slot == 18) vc += __shfl_xor(vc, 16, 64);
                if (slot == 20) vc += __shfl_xor(vc, 32, 64);
                if (slot == 22) { if (g == 0) S.w[pk(col)] = fma(2.0, vc, vown); }
                if (slot < 14 && (slot & 1) == 0) CC[(slot >> 1) + 1] = prF(80, (slot >> 1) + 1);
                if (slot < 28) {   // operand pair k + 1, four slots ahead of its MFMAs
                    const int k = slot >> 2;
                    if ((slot & 3) == 0) T0[k + 1] = prF(c, k + 1);
                    if ((slot & 3) == 1) T1[k + 1] = prF(16 + c, k + 1);
                }
                if (slot == 28) T2[0] = prF(32 + c, 0);
                if (slot == 29) T3[0] = prF(48 + c, 0);
            };
            if (!RIC_SPREAD) {
#pragma unroll
                for (int k = 0; k < 32; ++k) side(k);
            }
            d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[e >> 2][e & 3], T0[e >> 1][e & 1], a0, 0, 0, 0);
                if (RIC_SPREAD) {
                    side(2 * e);
                    if (e == 1) back_sub(3);
                    if (e == 4) back_sub(2);
                    if (e == 7) back_sub(1);
                    if (e == 10) gains_to_lds();
                    if (e == 11) out_piece(20, tn, have_out);
                    __builtin_amdgcn_sched_barrier(0);
                }
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[e >> 2][e & 3], T1[e >> 1][e & 1], a1, 0, 0, 0);
                if (RIC_SPREAD) { piece(e); side(2 * e + 1); __builtin_amdgcn_sched_barrier(0); }   // (the odd slots are the lighter ones)
            }
            pa = a0;
            pb = a1;
        }
        {
            // tiles 2, 3; in their shadow: the B block column's operands, the first two tiles' way to LDS
            auto side = [&](int slot) {
                if (slot < 28) {
                    const int k = slot >> 2;
                    if ((slot & 3) == 0) T2[k + 1] = prF(32 + c, k + 1);
                    if ((slot & 3) == 1) T3[k + 1] = prF(48 + c, k + 1);
                }
                if ((slot & 3) == 2) TB[slot >> 2] = prF(64 + c, slot >> 2);
                if (slot == 3) put_P(pa, 0);
                if (slot == 7) put_P(pb, 1);
            };
            if (!RIC_SPREAD) {
#pragma unroll
                for (int k = 0; k < 32; ++k) side(k);
            }
            d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[e >> 2][e & 3], T2[e >> 1][e & 1], a0, 0, 0, 0);
                if (RIC_SPREAD) { side(2 * e); out_piece(e, tn, have_out); __builtin_amdgcn_sched_barrier(0); }
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[e >> 2][e & 3], T3[e >> 1][e & 1], a1, 0, 0, 0);
                if (RIC_SPREAD) { if (16 + e < 30) piece(16 + e); side(2 * e + 1); __builtin_amdgcn_sched_barrier(0); }
            }
            pa = a0;
            pb = a1;
        }
        {
            // the B block column: two half-chains, added at the end; in their shadow: tiles 2, 3 to LDS and this wave's
            // 16 columns of Cxx_t (requested in Ph3 of the previous step) into the block's image
            d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[e >> 2][e & 3], TB[e >> 1][e & 1], a0, 0, 0, 0);
                if (RIC_SPREAD) {
                    if (e == 1) put_P(pa, 2);
                    if (e >= 2) {
                        S.X[(16 * w + 2 * (e - 2)) * kLdk + pk(lane)] = xs[2 * (e - 2)];
                        S.X[(16 * w + 2 * (e - 2) + 1) * kLdk + pk(lane)] = xs[2 * (e - 2) + 1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[(e + 8) >> 2][(e + 8) & 3], TB[(e + 8) >> 1][e & 1], a1, 0, 0, 0);
                if (RIC_SPREAD) {
                    if (e == 1) put_P(pb, 3);
                    if (e >= 6) {
                        S.X[(16 * w + 12 + 2 * (e - 6)) * kLdk + pk(lane)] = xs[12 + 2 * (e - 6)];
                        S.X[(16 * w + 12 + 2 * (e - 6) + 1) * kLdk + pk(lane)] = xs[12 + 2 * (e - 6) + 1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!RIC_SPREAD) {
                put_P(pa, 2);
                put_P(pb, 3);
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) S.X[(16 * w + jj) * kLdk + pk(lane)] = xs[jj];
            }
            Pw4 = a0 + a1;
        }
        RIC_TICK(0)
        __syncthreads();
        RIC_TICK(1)

        // ================= Ph2: Q(:, w) = C(:, w) + F' P(:, w), Quu part, q vectors ===========================
        // As in Ph1, everything that is not a matrix-core instruction rides in their shadows, a piece a slot: the symmetry
        // check of this wave's Cxx tiles (each element against its mirror: |C_ij - C_ji| <= 1e-13 max |C|), the operand
        // reads of the later tiles and [qx; qu] = [cx; cu] + F'w (riccati.py:112-116: two lanes a column, 32 products
        // each; the lanes beyond the 160 that have a column run along on the last one).  In front of / behind the MFMAs
        // they cost the phase ~1.2k cycles a step.
        // The operands themselves are streamed: a lane's 16 values of a tile column are eight 16-byte LDS reads, and read
        // whole in front of the phase (qi, P, two tiles of F': 32 reads a wave, 128 KB a CU at 128 B a cycle) they held the
        // first MFMA back ~1k cycles; now each pair is read a few slots before the MFMAs that use it.
        double dmax_t;
        {
            // pair i (elements 2 i, 2 i + 1) of this lane's operands of column `colx` of an image - ld16() by pieces
            auto pr = [&](const double* img, int colx, int i) {
                return *reinterpret_cast<const r64_d2*>(img + colx * kLdk + 16 * g + 2 * i);
            };
            r64_d2 q2[8], Pc2[8], A0[8], A1[8], B0[8], B1[8], C0[8];
            // in front: what the first MFMA of each chain needs - the tiles' starting values, pair 0 of the operands
#pragma unroll
            for (int i = 0; i < 4; ++i) q2[i] = pr(S.X, col, i);   // elements (16 I + g + 4 r, col), I = 0, 1
            Pc2[0] = pr(S.P, 16 * w + c, 0);
            A0[0] = pr(S.F, c, 0);
            A1[0] = pr(S.F, 16 + c, 0);
            Q[0] = d4{q2[0][0], q2[0][1], q2[1][0], q2[1][1]};
            Q[1] = d4{q2[2][0], q2[2][1], q2[3][0], q2[3][1]};
            double dm = 0.0, cm = 0.0, poison = 0.0, mi0 = 0.0, mi1 = 0.0;
            auto mirror = [&](int e) { return S.X[(16 * (e >> 2) + g + 4 * (e & 3)) * kLdk + pk(col)]; };   // (col, 16 I + g + 4 r)
            auto sym = [&](int e, double mi) {   // sym_lane(), one element
                const double d = q2[e >> 1][e & 1] - mi;
                dm = fmax(dm, fabs(d));
                cm = fmax(cm, fabs(mi));
                poison = fma(d, 0.0, poison);
                // (an IR pass otherwise sinks these to their first use behind the phase, out of the slot they were put in)
                asm volatile("" : "+v"(dm), "+v"(cm), "+v"(poison));
            };
            const int qcol = (tid >> 1) < 80 ? (tid >> 1) : 79, half = tid & 1;
            const r64_d2* fcol = reinterpret_cast<const r64_d2*>(&S.F[qcol * kLdk + 32 * half]);
            const r64_d2* wv = reinterpret_cast<const r64_d2*>(&S.w[32 * half]);
            double s0 = 0.0, s1 = 0.0;
            r64_d2 gf = {0.0, 0.0}, gw = {0.0, 0.0};
            r64_d2 fq2[2];   // k = 16 w + 4 s + g of B's column c: position 16 g + 4 w + s
            auto sideA = [&](int slot) {   // 32 slots under Q[0], Q[1]; MFMA pair e = slot / 2 uses operand pair e / 2
                // mirrors: read in slot e (two at a time in flight), compared two slots later
                if (slot >= 2 && slot < 18) sym(slot - 2, (slot & 1) ? mi1 : mi0);
                if (slot < 16) { if (slot & 1) mi1 = mirror(slot); else mi0 = mirror(slot); }
                if (slot < 4) q2[4 + slot] = pr(S.X, col, 4 + slot);   // the starting values of tiles 2, 3
                if (slot < 28) {   // operand pair k + 1, four slots ahead of its MFMAs
                    const int k = slot >> 2;
                    if ((slot & 3) == 0) Pc2[k + 1] = pr(S.P, 16 * w + c, k + 1);
                    if ((slot & 3) == 1) A0[k + 1] = pr(S.F, c, k + 1);
                    if ((slot & 3) == 2) A1[k + 1] = pr(S.F, 16 + c, k + 1);
                }
                if (slot == 28) B0[0] = pr(S.F, 32 + c, 0);
                if (slot == 29) B1[0] = pr(S.F, 48 + c, 0);
                if (slot >= 20 && slot < 26) {   // max |C| over the wave
                    const int o = 32 >> (slot - 20);
                    cm = fmax(cm, __shfl_xor(cm, o, 64));
                }
                if (slot == 27) { if (lane == 0) S.cmax[w] = cm; }
            };
            auto sideB = [&](int slot) {   // 32 slots under Q[2], Q[3]
                if (slot < 28) {
                    const int k = slot >> 2;
                    if ((slot & 3) == 0) B0[k + 1] = pr(S.F, 32 + c, k + 1);
                    if ((slot & 3) == 1) B1[k + 1] = pr(S.F, 48 + c, k + 1);
                }
                if ((slot & 3) == 2) C0[slot >> 2] = pr(S.F, 64 + c, slot >> 2);   // the Qux tile's operands, all eight pairs
                if (slot == 3 || slot == 7) fq2[slot >> 2] = *reinterpret_cast<const r64_d2*>(&S.F[(64 + c) * kLdk + 16 * g + 4 * w + 2 * (slot >> 2)]);
                // the q vectors: pair k of 16 read in slot 8 + k and multiplied a slot later
                if (slot >= 9 && slot < 25) { s0 = fma(gf[0], gw[0], s0); s1 = fma(gf[1], gw[1], s1); }
                if (slot >= 8 && slot < 24) { gf = fcol[slot - 8]; gw = wv[slot - 8]; }
            };
            if (!RIC_SPREAD) {
#pragma unroll
                for (int k = 0; k < 32; ++k) sideA(k);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {   // two independent accumulator chains at a time (see Ph1)
                Q[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0[e >> 1][e & 1], Pc2[e >> 1][e & 1], Q[0], 0, 0, 0);
                if (RIC_SPREAD) { sideA(2 * e); __builtin_amdgcn_sched_barrier(0); }
                Q[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A1[e >> 1][e & 1], Pc2[e >> 1][e & 1], Q[1], 0, 0, 0);
                if (RIC_SPREAD) { sideA(2 * e + 1); __builtin_amdgcn_sched_barrier(0); }
            }
            dmax_t = dm + poison;
            Q[2] = d4{q2[4][0], q2[4][1], q2[5][0], q2[5][1]};
            Q[3] = d4{q2[6][0], q2[6][1], q2[7][0], q2[7][1]};
            if (!RIC_SPREAD) {
#pragma unroll
                for (int k = 0; k < 32; ++k) sideB(k);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                Q[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(B0[e >> 1][e & 1], Pc2[e >> 1][e & 1], Q[2], 0, 0, 0);
                if (RIC_SPREAD) { sideB(2 * e); __builtin_amdgcn_sched_barrier(0); }
                Q[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(B1[e >> 1][e & 1], Pc2[e >> 1][e & 1], Q[3], 0, 0, 0);
                if (RIC_SPREAD) { sideB(2 * e + 1); __builtin_amdgcn_sched_barrier(0); }
            }
            double sum = s0 + s1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                Q[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(C0[e >> 1][e & 1], Pc2[e >> 1][e & 1], Q[4], 0, 0, 0);
                if (RIC_SPREAD) {
                    if (e >= 4 && e < 8) out_piece(16 + (e - 4), tn, have_out);   // last step's gains (in LDS since Ph1)
                    if (e == 8) out_piece(21, tn, have_out);
                    if (e == 0) sum += __shfl_xor(sum, 1, 64);
                    if (e == 2) {
                        if (tid < 160 && half == 0) {
                            if (qcol < 64) S.qx[qcol] = cq + sum;
                            else S.qu[qcol - 64] = cq + sum;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((e & 3) == 3) {
                    qp = __builtin_amdgcn_mfma_f64_16x16x4f64(fq2[e >> 3][(e >> 2) & 1], Pw4[e >> 2], qp, 0, 0, 0);
                    if (RIC_SPREAD) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!RIC_SPREAD) {
#pragma unroll
                for (int j = 16; j < 20; ++j) out_piece(j, tn, have_out);
                out_piece(21, tn, have_out);
                sum += __shfl_xor(sum, 1, 64);
                if (tid < 160 && half == 0) {
                    if (qcol < 64) S.qx[qcol] = cq + sum;
                    else S.qu[qcol - 64] = cq + sum;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            S.QuuP[w][(g + 4 * r) + kLd16 * c] = qp[r];   // (the Qux tile Q[4] stays in this wave's registers)
        }
        RIC_TICK(2)
        __syncthreads();
        RIC_TICK(3)

        // ================= Ph3: block elimination on the matrix cores (every wave: Quu redundantly, its own Qux tile) ===
        {
            const double tol = 1e-13 * fmax(fmax(S.cmax[0], S.cmax[1]), fmax(S.cmax[2], S.cmax[3]));
            if (!(dmax_t <= tol)) S.asym = 1;   // read by everybody behind the next barrier
        }
        // The F image of this step is dead: step t - 1 moves in, and the slice of Cxx for it is requested (unconditional: the
        // last one is a harmless re-load) - both in pieces, inside the waits of the elimination below (stage_chunk() behind
        // the broadcast read of a pivot block, fetch_X_chunk() behind its first three MFMAs); in front of the chain they cost it
        // ~500 cycles.
        const unsigned tq = (unsigned)__builtin_amdgcn_readfirstlane(t >= 1 ? t - 1 : 0);
        auto stage_chunk = [&](int i) {   // 21 LDS stores in three chunks
#pragma unroll
            for (int s7 = 0; s7 < 7; ++s7) {
                const int q = 7 * i + s7;
                if (q < 16) S.F[offA[q]] = fA[q];
                else if (q < 20) S.F[offB[q - 16]] = fB[q - 16];
                else S.F[offc] = fc;
            }
        };
        auto fetch_X_chunk = [&](int i) {   // 16 loads in four chunks
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int jj = 4 * i + j4;
                xs[jj] = r64_ld(rCxx, (FULL || (16 * wu + jj) < n) ? voRow : kR64NoStore, (tq * (unsigned)nn + (unsigned)(n * (16 * wu + jj))) * 8u);
            }
        };
        // Quu = L D L' with 4 x 4 blocks (L unit lower BLOCK triangular, D block diagonal), no pivoting: a non-positive
        // scalar pivot inside a block flags the problem for the generic kernel.  In the matrix cores' D layout register
        // r of a 16 x 16 tile holds rows 4 r + g - i.e. ROW BLOCK r - and, for a symmetric tile, that register is at
        // the same time the B operand "rows of block r" and the A operand "columns of block r".  Step k:
        //   Pi = (Quu^(k)(blk, blk))^-1            4 x 4, every lane redundantly from a broadcast through LDS
        //   X' = Pi Quu^(k)(blk, :), Z_k = Pi R(blk, :), zq_k = Pi rq(blk)     one instruction each (Pi zero-padded)
        //   Quu^(k+1) = Quu^(k) - Quu(:, blk) X',  R -= Quu(:, blk) Z_k,  rq -= Quu(:, blk) zq_k   (rows below the block)
        // after which R = Y = L^-1 Qux(:, w) and the rows X' are the unit upper factor U = L'.  Ph4 needs only Y and Z;
        // the gains come from the block back substitution K = -U^-1 Z (three more instructions per tile).
        bool ok = true;
        d4 Sd, Rq = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = g + 4 * r;
            double x = 0.0;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) x += S.QuuP[ww][row + kLd16 * c];
            Sd[r] = (row < m && c < m) ? x : (row == c ? 1.0 : 0.0);   // identity padding beyond du
            Rq[r] = (c == 0 && row < m) ? S.qu[row] : 0.0;             // qu rides along as column 0 of a tile of its own
        }
        d4& R = Q[4];   // this wave's Qux tile, rows i = g + 4 r, column 16 w + c
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // the pivot block to every lane: element (row 4 k + g, column c) is register k of lane (c, g), c in the block
            if ((c >> 2) == k) S.piv[w][g + 4 * (c & 3)] = Sd[k];
            wave_fence();
            double P[16];
            {
                const r64_d2* src = reinterpret_cast<const r64_d2*>(S.piv[w]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const r64_d2 x2 = src[e];
                    P[2 * e] = x2[0];
                    P[2 * e + 1] = x2[1];
                }
            }
            wave_fence();
            if (k < 3) {
                __builtin_amdgcn_sched_barrier(0);
                stage_chunk(k);
                __builtin_amdgcn_sched_barrier(0);
            }
            // Pi = P^-1 by 2 x 2 blocks, P = [A B'; B C] (lower triangle of P used, P(a, b) at a + 4 b):
            //   A^-1 = adj(A) / det A,  S = C - B A^-1 B',  S^-1 = adj(S) / det S,
            //   Pi = [A^-1 + G' S^-1 G, -G' S^-1; -S^-1 G, S^-1],  G = B A^-1
            // - half the dependent chain of a 4 x 4 L D L' (two reciprocals in sequence instead of four).  P is positive
            // definite iff a00, det A, s00 and det S are positive (leading minors of P: a00, det A, det A s00, det A det S).
            const double a00 = P[0], a10 = P[1], a11 = P[5];
            const double b00 = P[2], b10 = P[3], b01 = P[6], b11 = P[7];      // B = P(2..3, 0..1): B(i, j) = P[2 + i + 4 j]
            const double c00 = P[10], c10 = P[11], c11 = P[15];
            const double detA = fma(a00, a11, -(a10 * a10));
            const double iA = rcp_nr(detA);
            const double ia00 = a11 * iA, ia10 = -a10 * iA, ia11 = a00 * iA;    // A^-1 (symmetric)
            // G = B A^-1 (2 x 2)
            const double g00 = fma(b00, ia00, b01 * ia10), g01 = fma(b00, ia10, b01 * ia11);
            const double g10 = fma(b10, ia00, b11 * ia10), g11 = fma(b10, ia10, b11 * ia11);
            // S = C - G B'
            const double s00 = c00 - fma(g00, b00, g01 * b01);
            const double s10 = c10 - fma(g10, b00, g11 * b01);
            const double s11 = c11 - fma(g10, b10, g11 * b11);
            const double detS = fma(s00, s11, -(s10 * s10));
            const double iS = rcp_nr(detS);
            const double p22 = s11 * iS, p32 = -s10 * iS, p33 = s00 * iS;       // S^-1
            ok = ok && (a00 > 0.0) && (detA > 0.0) && (s00 > 0.0) && (detS > 0.0);
            // lower-left block -S^-1 G: rows 2..3, columns 0..1
            const double p20 = -fma(p22, g00, p32 * g10), p21 = -fma(p22, g01, p32 * g11);
            const double p30 = -fma(p32, g00, p33 * g10), p31 = -fma(p32, g01, p33 * g11);
            // upper-left block A^-1 - G' (lower-left)
            const double p00 = ia00 - fma(g00, p20, g10 * p30);
            const double p10 = ia10 - fma(g01, p20, g11 * p30);
            const double p11 = ia11 - fma(g01, p21, g11 * p31);
            // the A operand "Pi zero-padded to 16 x 4": A[row c][k-index g] = Pi(c, g) for c < 4 - per-lane 0 / 1 masks
            double Aop = mk[7] * p22, Aop2 = mk[0] * p00;   // (two chains; the late entries last)
            Aop = fma(mk[8], p32, Aop); Aop2 = fma(mk[1], p10, Aop2);
            Aop = fma(mk[9], p33, Aop); Aop2 = fma(mk[4], p11, Aop2);
            Aop = fma(mk[2], p20, Aop); Aop2 = fma(mk[3], p30, Aop2);
            Aop = fma(mk[5], p21, Aop); Aop2 = fma(mk[6], p31, Aop2);
            Aop += Aop2;
            const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
            const d4 Xs = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop, Sd[k], zero4, 0, 0, 0);   // rows 0..3: X' = Pi Quu(blk, :)
            const d4 Zr = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop, R[k], zero4, 0, 0, 0);    // rows 0..3: Z_k = Pi Y_k
            const d4 Zv = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop, Rq[k], zero4, 0, 0, 0);   // rows 0..3, column 0: zq_k
            __builtin_amdgcn_sched_barrier(0);
            fetch_X_chunk(k);
            __builtin_amdgcn_sched_barrier(0);
            // columns of the block, rows below it (by symmetry register k again), negated: the rank-4 downdates (below the
            // last block there is nothing)
            if (k < 3) {
                const double nA = (c >= 4 * (k + 1)) ? -Sd[k] : 0.0;
                Sd = __builtin_amdgcn_mfma_f64_16x16x4f64(nA, Xs[0], Sd, 0, 0, 0);
                R = __builtin_amdgcn_mfma_f64_16x16x4f64(nA, Zr[0], R, 0, 0, 0);
                Rq = __builtin_amdgcn_mfma_f64_16x16x4f64(nA, Zv[0], Rq, 0, 0, 0);
            }
            Zt[k] = Zr[0];
            Zq4[k] = Zv[0];
            S.U[w][(4 * k + g) + kLd16 * c] = Xs[0];           // row 4 k + g of U (its entries right of the block are used)
            S.K[(4 * k + g) + kLd16 * col] = -Zr[0];          // -Z: the A operand of the V update
        }
        if (!ok) {  // identical in every wave: block-uniform exit
            bad = t;
            break;
        }
        // zq = D^-1 L^-1 qu is what v needs (Ph4); the back substitution that turns Z, zq into the gains is not on the
        // value recursion's path: it rides in the next step's Ph1 (back_sub() there)
        if (w == 1 && c == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S.zq[g + 4 * r] = Zq4[r];
        }
        RIC_TICK(4)
        __syncthreads();
        RIC_TICK(5)

        // ================= Ph4: V(:, w) = Qxx(:, w) - Z' Y(:, w), v = qx - Y' zq, outputs (S.K holds -Z, Q[4] holds Y) ====
        if (S.asym != 0) {   // this step's Cxx is not symmetric (block-uniform): nothing of the step is stored
            bad = t;
            break;
        }
        {
            // all sixteen A operands (-Z) first: read one at a time in front of its MFMA, each exposed the LDS latency
            double ka[16];
#pragma unroll
            for (int I = 0; I < 4; ++I)
#pragma unroll
                for (int s = 0; s < 4; ++s) ka[4 * I + s] = S.K[(4 * s + g) + kLd16 * (16 * I + c)];
            const double zq0 = S.zq[g], zq1 = S.zq[g + 4], zq2 = S.zq[g + 8], zq3 = S.zq[g + 12], qxc = S.qx[colok ? col : 0];
            __builtin_amdgcn_sched_barrier(0);
            // four independent accumulators: s outermost, so that no MFMA waits for the one before it
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int I = 0; I < 4; ++I)
                    Q[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[4 * I + s], Q[4][s], Q[I], 0, 0, 0);
            double part = fma(Q[4][0], zq0, 0.0);   // (Y' zq)[col]
            part = fma(Q[4][1], zq1, part);
            part = fma(Q[4][2], zq2, part);
            part = fma(Q[4][3], zq3, part);
            part = xor_sum16(part);
            vown = colok ? qxc - part : 0.0;  // v = qx + 2 Qux' kff = qx - Qux' Quu^-1 qu   (riccati.py:124)
        }
#pragma unroll
        for (int I = 0; I < 4; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) Vt[I][r] = ((okbits >> (4 * I + r)) & 1u) ? Q[I][r] : 0.0;  // padding stays exactly zero
        // (the step's outputs - V, v, K, kff - are stored from the shadows of the NEXT step's Ph1: out_piece())
        RIC_TICK(6)
    }
    if (bad < 0) {   // the outputs of step 0 (block-uniform)
        back_sub(3);
        back_sub(2);
        back_sub(1);
        gains_to_lds();
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kOutPieces; ++j) out_piece(j, 0u, true);
    }
#if RIC_DIAG
    if (tid == 0)
        for (int k = 0; k < 7; ++k) ov[k] = (double)tick_[k];
#endif
    // a non-positive pivot: Quu is singular or indefinite - the generic kernel's pivoted LU tells which (riccati.py:118)
    if (tid == 0) {
        if (a.status) a.status[p] = bad >= 0 ? kRicRedo : 0;
        else if (bad >= 0) a.redo[p] = 1;
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
    if ((double)n * n * (double)(T + 1) * 8.0 >= 4294967296.0 - 1024.0) return TRAJOPT_OK;   // 32-bit offsets inside a problem's slices
    if (NP > kRicMaxProblems) return TRAJOPT_OK;   // (more problems than flags: the generic kernel takes them all)
    int32_t* redo = nullptr;
    if (!status) {   // no caller-owned flag memory: the per-device static array (one such call per GPU at a time)
        const int rc = ric_flags(&redo);
        if (rc != TRAJOPT_OK) return rc;
        TRAJOPT_HIP_CHECK(hipMemsetAsync(redo, 0, (size_t)NP * sizeof(int32_t), st));
    }
    *handled = true;
    Ric64Args a{Cxx, cx, Cuu, cu, Cxu, A, B, c, V, v, K, kff, status, redo, n, m, T};
    const size_t lds = sizeof(Ric64Lds);
    if (n == 64 && m == 16) {
        TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)ric_mfma64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ric_mfma64_kernel<true>, dim3((unsigned)NP), dim3(256), lds, st, a);
    } else {
        TRAJOPT_HIP_CHECK(hipFuncSetAttribute((const void*)ric_mfma64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ric_mfma64_kernel<false>, dim3((unsigned)NP), dim3(256), lds, st, a);
    }
    TRAJOPT_HIP_CHECK(hipGetLastError());
    // the flagged problems, by the kernel without those assumptions (a workgroup of an unflagged problem leaves at once)
    return launch_riccati_wg(Cxx, cx, Cuu, cu, Cxu, A, B, c, n, m, T, NP, TRAJOPT_LAYOUT_PROBLEM_MAJOR, V, v, K, kff, status, st, status ? status : redo);
}

}  // namespace trajopt
