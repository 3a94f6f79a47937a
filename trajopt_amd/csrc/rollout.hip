// Forward rollouts (row A3) and single env steps (row A8).
//
//   iLQR.forward_pass     /root/reference/trajopt/ilqr/ilqr.py:89-102
//   LinearControl.action  /root/reference/trajopt/ilqr/objects.py:137-139
//   env.dynamics / cost   /root/reference/trajopt/envs/cartpole/cartpole.py:62-97,112-127 (and siblings)
//
// One thread integrates one (problem, alpha) pair over the whole horizon; the thread index runs over
// problems fastest, so in PROBLEM_MINOR layout every load of K/kff/xref/uref and every store of the
// candidate state/action/cost is one coalesced row per wave, and the alphas of a problem re-read the
// same gain rows from L2.  The inputs of step t+1 are fetched while step t integrates.
#include <cstddef>
#include <type_traits>
#include "sys.h"

namespace trajopt {

struct RolloutArgs {
    trajopt_env_t env;
    const double *x0, *K, *kff, *xref, *uref, *weighting;
    const int32_t* active;
    double *state, *action, *cost, *ret;
    int32_t* finite;
    double alphas[TRAJOPT_MAX_ALPHAS];
    size_t NP;
    int T, nb_alphas;
    // optional restriction (staged line search of the driver): alphas [alpha_begin, nb_alphas) only, and, when
    // plist is given, only the *pcount problems listed there (compacted, so that waves stay full)
    int alpha_begin;
    const int32_t* plist;
    const int32_t* pcount;
    // Driver only (RolloutPlanes in common.h): state / action are the driver's TIME-BLOCKED candidate planes; alpha ai
    // goes to plane first_plane + ai, the lanes of stopped problems to junk_plane.
    int first_plane, junk_plane;
    int xcd_tiles;   // 1: workgroup -> (tile of 64 problems, alpha) mapping that keeps a tile's alphas on one XCD
    // 1: thread index = problem * nA + alpha (the alphas of a problem in neighbouring lanes) instead of alpha * NP +
    // problem.  For PROBLEM_MAJOR gains (dx > 6): a lane reads its problem's K_t as dx du consecutive doubles, so 64
    // different problems in a wave are 64 x 6 cache lines per step (quadrotor: 48 KB, more than the CU's vector cache
    // holds - every one of the 52 loads of a step went to L2 for 64 lines); with the alphas of a problem side by
    // side a wave touches an eighth of that and the other lanes hit the same lines.
    int alpha_minor;
    // ... and the gains K / kff hold TRAJOPT_GAIN_BANKS banks, of which problem p reads bank out_bank[p] (layout:
    // trajopt_backward_io_t::out_bank); NULL: plain arrays
    const int32_t* out_bank;
    // Second pass of the driver's two-pass line search: ONE lane per problem, which rolls out alpha sel[p] again (the
    // step size select_kernel accepted; < 0: nothing to do) and writes its trajectory to plane first_plane; NULL: every
    // (problem, alpha)
    const int32_t* sel;
};

// index into the gains K / kff, which hold kmul banks when the driver multi-buffers them
template <bool KMINOR>
__device__ __forceinline__ size_t gk(size_t eps, size_t slices, int t, size_t e, size_t p, size_t NP, int kmul, int bank) {
    return KMINOR ? (eps * (size_t)(kmul * t + bank) + e) * NP + p
                  : ((size_t)bank * NP + p) * eps * slices + eps * (size_t)t + e;
}

template <bool MINOR>
__device__ __forceinline__ size_t gi(size_t e_total, size_t E_total, size_t p, size_t NP) {
    return MINOR ? e_total * NP + p : p * E_total + e_total;
}

// Staging of a wave's candidate stores for the time-blocked planes (TB; common.h).  A lane produces one double per
// element and step, but a plane wants the 8 steps of a block next to each other: written as they come, every store
// instruction would touch 32 cache lines with 16 bytes each (measured: the rollout took twice as long).  So the wave
// collects a block - 8 steps x 64 lanes per element - in LDS, the 8 steps of a lane next to each other, and writes
// it out 16 bytes a lane: store k of a row covers the lane-problems 16 k .. 16 k + 15 with their 8 steps each, 1 KB
// contiguous when the wave's problems are neighbours in one bank (and whole 64-byte runs otherwise) - half as many
// LDS reads and store instructions as 8 bytes a lane.  A lane's 8 steps are padded to 10 doubles (16-byte aligned
// reads; the puts of a step are then a 2-way bank conflict, the cheapest padding that keeps the alignment).
constexpr int kStagePad = 10;
#ifndef ROLL_DIAG
#define ROLL_DIAG 0   // timing experiments only (wrong results): 1 = no write-out of the staged blocks, 2 = no staging either
#endif
typedef double stage_v2 __attribute__((ext_vector_type(2)));
// Models with more than 8 rows a step (quadrotor: 16) stage HALF blocks: 4 steps x 64 lanes per element, unpadded (32 KB
// a wave at 16 rows, where whole padded blocks would take 82 KB - one wave per CU), written out as 32-byte half cells,
// which are whole sectors.  Round 2 wrote those models' candidates straight from the lanes instead, 8 bytes a lane and 64
// cells a store: the 16 stores of a step were 70 % of the quadrotor's rollout (3.27 ms with them, 1.00 ms without).
constexpr int kHalfSteps = 4;
template <int ROWS, int PAD = kStagePad>
struct alignas(16) TbStage {
    double v[ROWS * 64 * PAD];
    unsigned long long bx[64], bu[64];   // per lane-problem: element offset of (block 0, row 0) in state / action
};
// rows [row0, row0 + rows) of the stage = block `blk` of `base` (a plane set with `rows` rows per block)
// Every store is an unconditional instruction (a store under a branch cannot be counted by hipcc's waitcnt pass, which
// then makes the loads behind it wait for all of them): lanes without a result of their own - beyond the batch, or a
// stopped problem - write what they computed along with the others, into planes nobody reads (see the kernel).
template <int ROWS>
__device__ __forceinline__ void tb_flush(const TbStage<ROWS>& sg, const unsigned long long* off, double* base, int row0,
                                         int rows, int blk, size_t NP, unsigned lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned sp = lane & 3u, pj = lane >> 2;   // the lane's pair of steps, its problem among the store's 16
    const size_t rstride = NP * kTB;
    double* dst0 = base + ((size_t)blk * rows) * rstride + 2 * sp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned prob = (unsigned)k * 16 + pj;
        double* d = dst0 + off[prob];
        for (int r = 0; r < rows; ++r)
            *reinterpret_cast<stage_v2*>(d + (size_t)r * rstride) =
                *reinterpret_cast<const stage_v2*>(&sg.v[((row0 + r) * 64 + prob) * kStagePad + 2 * sp]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Half blocks (TbStage<ROWS, kHalfSteps>): rows [row0, row0 + rows) of the stage = steps 4 hb .. 4 hb + 3 of `base`.  A store
// covers 32 lane-problems with their 4 steps each (16 bytes a lane, two lanes a half cell).
template <int ROWS>
__device__ __forceinline__ void tb_flush_half(const TbStage<ROWS, kHalfSteps>& sg, const unsigned long long* off, double* base,
                                              int row0, int rows, int hb, size_t NP, unsigned lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned sp = lane & 1u, pj = lane >> 1;
    const size_t rstride = NP * kTB;
    double* dst0 = base + ((size_t)(hb >> 1) * rows) * rstride + (size_t)(hb & 1) * kHalfSteps + 2 * sp;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const unsigned prob = (unsigned)k * 32 + pj;
        double* d = dst0 + off[prob];
        for (int r = 0; r < rows; ++r)
            *reinterpret_cast<stage_v2*>(d + (size_t)r * rstride) =
                *reinterpret_cast<const stage_v2*>(&sg.v[((row0 + r) * 64 + prob) * kHalfSteps + 2 * sp]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// MINOR: layout of x0 / xref / uref / candidates;  KMINOR: layout of the gains K, kff (the batched driver keeps
// them PROBLEM_MAJOR for the matrix-core backward kernel when dx > 6).
// FAST: the batched driver's call - state and action planes given, no per-step cost plane, plain cost (no slew rate,
// no angle wrapping, no dt scaling) - with the stores unconditional: a store under a (wave-uniform) `if (pointer)` makes the waitcnt pass wait for EVERY outstanding
// memory operation (vmcnt(0)) before the prefetched gains are used, i.e. for this step's stores to be acknowledged.
// TB: state / action are the driver's time-blocked candidate planes, written through the LDS stage above.
// UNR (with FAST and TB): the 8 steps of a block are straight-line code.  That is the fastest form for a lone wave per
// SIMD (no wait at all for a block's write-out), but it takes 256 registers and more, i.e. one wave per SIMD; batches
// that oversubscribe the chip take the rolled form (two waves per SIMD hide its one wait per block).
// CO (with FAST, without TB): returns only, no trajectory is written - the first pass of the two-pass line search.
// (The returns-only cartpole takes 176 registers, 2 waves per SIMD; held to 168 for 3 - __launch_bounds__(64, 3), 36
// bytes of spills - 65 536 cartpoles took 8.4 instead of 7.8 ms an iteration, 16 384 2.98 instead of 2.76.)
// KL (with FAST, TB, PROBLEM_MAJOR gains, dx + du > 8, the alphas of a problem in neighbouring lanes, >= 8 alphas): the
// gains K_t reach the lanes THROUGH LDS.  A wave then holds at most 8 different problems, i.e. 8 different K_t, but
// lane by lane that was 64 copies of dx du doubles (quadrotor: 96 registers, 48 load instructions a step, and with the
// dynamics next to them 512 registers were not enough: the set went through AGPRs and ~30 doubles through scratch,
// each behind a full wait).  Here lane l fetches an eighth of the K_t of the wave's problem l / 8 a step ahead (dx du / 8
// doubles, 16 bytes a load), the wave writes the 8 matrices into its LDS image at the top of the step and every lane
// reads its problem's rows from there as the control law consumes them (slots padded by 2 doubles: the 8 rows a wave
// reads at once fall into different banks).  Same products in the same order: results are bitwise those of KL = false.
template <class SYS, bool MINOR, bool KMINOR, bool FAST, bool TB, bool UNR, bool CO = false, bool KL = false>
__global__ __launch_bounds__(64) void rollout_kernel(RolloutArgs a) {
    constexpr int N = SYS::N, M = SYS::M;
    static_assert(!KL || (FAST && TB && !KMINOR && !UNR && !CO && (M * N) % 16 == 0), "KL: see above");
    constexpr int kKSlot = M * N + 2;   // doubles per problem in the LDS image of the gains
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x;
    const size_t NP = a.NP;
    size_t p;
    int ai;
    bool idle = false;  // no early return before the barrier below
    if (a.sel) {
        p = idx < NP ? idx : 0;
        const int s = idx < NP ? a.sel[p] : -1;
        idle = s < 0;
        ai = idle ? a.alpha_begin : s;
    } else if (a.xcd_tiles) {
        // The driver's order: the nA step sizes of a tile of 64 problems are consecutive workgroups OF ONE XCD
        // (workgroups b and b + 8 share one, cdna guide): they run at about the same time and read the tile's gains
        // and reference through one L2, where the plain order - alpha slowest - fetches them from HBM once per alpha.
        const unsigned nA = (unsigned)(a.nb_alphas - a.alpha_begin);
        const unsigned b = blockIdx.x, j = b >> 3;
        const size_t q = ((size_t)(j / nA) * 8 + (b & 7u)) * 64 + lane;   // index into the problems (or the list)
        const size_t cnt = a.plist ? (size_t)*a.pcount : NP;
        idle = q >= cnt;
        p = idle ? 0 : (a.plist ? (size_t)a.plist[q] : q);
        ai = a.alpha_begin + (int)(j % nA);
    } else if (a.plist) {
        const size_t cnt = (size_t)*a.pcount;
        idle = idx >= cnt * (size_t)(a.nb_alphas - a.alpha_begin);
        p = idle ? 0 : (size_t)a.plist[idx % (cnt ? cnt : 1)];
        ai = idle ? a.alpha_begin : a.alpha_begin + (int)(idx / (cnt ? cnt : 1));
    } else if (a.alpha_minor) {
        const size_t nA = (size_t)(a.nb_alphas - a.alpha_begin);
        idle = idx >= NP * nA;
        p = idle ? 0 : idx / nA;
        ai = idle ? a.alpha_begin : a.alpha_begin + (int)(idx % nA);
    } else {
        idle = idx >= NP * (size_t)(a.nb_alphas - a.alpha_begin);
        p = idle ? 0 : idx % NP;
        ai = idle ? a.alpha_begin : a.alpha_begin + (int)(idx / NP);
    }
    // The env constants (dt, limits, weights, physical parameters: ~30 doubles for a cartpole) plus the sincos
    // coefficients do not fit the scalar register file: left in the kernel-argument segment they are spilled and
    // re-read (s_load + wait, v_readlane) inside the time loop, which a lone wave pays for in full.  A copy read
    // back from LDS lives in VGPRs (plentiful here); only the fields a model uses survive as registers.  The LQR
    // env indexes its matrices with run-time dimensions and keeps the kernel-argument copy.
    __shared__ trajopt_env_t s_env;
    // Models with more than 8 rows a step (quadrotor: 16) stage half blocks (TbStage above).
    // (ROLL_HALF_SMALL = 1 gives the rolled forms of the small models the half-block stage too - double cartpole: 17 KB
    // instead of 39 KB of LDS and 232 registers, i.e. two waves per SIMD, so that the 1408 waves of 8192 x 11 alphas are
    // resident at once instead of going through the chip in two rounds.  Measured, A/B on one box: slower everywhere -
    // 8192 double cartpoles 1.55 -> 2.02 ms, 16 384: 2.6 -> 3.2, 7168 cartpoles 0.88 -> 1.12, 131 072 cartpoles 8.9 ->
    // 9.4: two waves of a vector-bound chain share one pipe, and the write-out is twice as many, half as wide stores.)
#ifndef ROLL_HALF_SMALL
#define ROLL_HALF_SMALL 0
#endif
    constexpr bool kHalf = TB && (N + M > 8 || (ROLL_HALF_SMALL && !UNR));
    constexpr int kSteps = kHalf ? kHalfSteps : kTB, kPad = kHalf ? kHalfSteps : kStagePad;
    __shared__ TbStage<TB ? N + M : 1, kPad> s_stage;
    typedef double kl_v2 __attribute__((ext_vector_type(2)));
    __shared__ kl_v2 s_gains[KL ? 8 * kKSlot / 2 : 1];
    if constexpr (SYS::kStatic) {
        // everything but the LQR matrices: [0, lin_A) and [phys, end) - 64 doubles, one pass of the wave
        constexpr unsigned kHead = offsetof(trajopt_env_t, lin_A) / sizeof(double);
        constexpr unsigned kPhys = offsetof(trajopt_env_t, phys) / sizeof(double);
        constexpr unsigned kEnd = sizeof(trajopt_env_t) / sizeof(double);
        double* dst = reinterpret_cast<double*>(&s_env);
        const double* src = reinterpret_cast<const double*>(&a.env);
        for (unsigned i = threadIdx.x; i < kHead + (kEnd - kPhys); i += blockDim.x) {
            const unsigned j = i < kHead ? i : kPhys + (i - kHead);
            dst[j] = src[j];
        }
        __syncthreads();
    }
    // A lane with nothing to do (beyond the batch, or a stopped problem) leaves at once - except under TB, where the
    // whole wave writes the staged blocks out together: there it runs along on its own problem's data (beyond the
    // batch: problem 0's) and writes its candidates like the others, but into a plane of their own that nobody
    // reads, and not its return; a wave without any live lane still leaves as a whole.
    const bool live = !idle && !(a.active && a.active[p] == 0);
    if constexpr (TB) {
        if (__builtin_amdgcn_ballot_w64(live) == 0) return;
    } else {
        if (!live) return;
    }
    trajopt_env_t ev;
    if constexpr (SYS::kStatic) {
        ev = s_env;
        ev.slew_rate = a.env.slew_rate;  // flags stay wave-uniform so that their branches stay scalar
        ev.periodic = a.env.periodic;
        ev.cost_times_dt = a.env.cost_times_dt;
        ev.nb_features = a.env.nb_features;
#pragma unroll
        for (int f = 0; f < TRAJOPT_ENV_MAX_STATE; ++f) ev.features[f] = a.env.features[f];
        // x + 0.0 is an arithmetic result, i.e. canonical: fmin / fmax against it need no v_max(x, x) quieting per use
#pragma unroll
        for (int j = 0; j < M; ++j) ev.umax[j] += 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) ev.xmax[i] += 0.0;
    }
    const trajopt_env_t& e = SYS::kStatic ? ev : a.env;
    const int n = SYS::n(e), m = SYS::m(e), T = a.T;
    // (second pass: a lane without an accepted step size runs along on alpha = 0, i.e. on the reference itself - with a
    // rejected step size its states may grow until the whole wave takes sincos' big-argument path)
    const double alpha = (a.sel && idle) ? 0.0 : a.alphas[ai];
    const size_t En1 = (size_t)n * (T + 1), En = (size_t)n * T, Em = (size_t)m * T, Enm = (size_t)n * m * T;
    const int kmul = a.out_bank ? TRAJOPT_GAIN_BANKS : 1, bank = gains_bank(a.out_bank, p);
    // candidate planes: alpha slowest (TB: inside this iteration's bank; a lane without a result of its own writes junk_plane)
    // (second pass of the two-pass line search: ONE plane for whatever alpha a problem accepted.  Letting the problems
    // that accepted none write their own slot of it instead of the junk plane - whole 1 KB runs - changed nothing:
    // 65 536 cartpoles 7.2 ms an iteration either way.)
    const size_t plane = TB ? (size_t)(live ? a.first_plane + (a.sel ? 0 : ai) : a.junk_plane) : (size_t)ai;
    double* st = (!TB && a.state) ? a.state + plane * En1 * NP : nullptr;
    double* ac = (!TB && a.action) ? a.action + plane * Em * NP : nullptr;
    double* co = a.cost ? a.cost + (size_t)ai * (size_t)(T + 1) * NP : nullptr;
    const size_t bx_own = TB ? plane * tb_plane(n, T + 1, NP) + p * kTB : 0, bu_own = TB ? plane * tb_plane(m, T, NP) + p * kTB : 0;
    if constexpr (TB) {
        s_stage.bx[lane] = bx_own;
        s_stage.bu[lane] = bu_own;
    }
    // TB: element i of state slice s / element j of action slice t goes to the stage.  Both blocks b = t / 8 are
    // written out together after the step with t % 8 == 7 - BEFORE that step's new state, slice 8 b + 8, which is slot
    // 0 of the next block, is put - and whatever the last, partial blocks hold after the last step.
    auto put_x = [&](int s, int i, double v) {
        if (ROLL_DIAG >= 2) return;
        s_stage.v[(i * 64 + lane) * kPad + (s & (kSteps - 1))] = v;
    };
    auto put_u = [&](int t, int j, double v) {
        if (ROLL_DIAG >= 2) return;
        s_stage.v[((n + j) * 64 + lane) * kPad + (t & (kSteps - 1))] = v;
    };
    // b: the block (of kSteps steps: 8, or 4 for the half-block stage) that is complete
    auto flush_block = [&](int b) {
        if (ROLL_DIAG >= 1) return;
        if constexpr (kHalf) {
            tb_flush_half(s_stage, s_stage.bx, a.state, 0, n, b, NP, lane);
            tb_flush_half(s_stage, s_stage.bu, a.action, n, m, b, NP, lane);
        } else if constexpr (TB) {
            tb_flush(s_stage, s_stage.bx, a.state, 0, n, b, NP, lane);
            tb_flush(s_stage, s_stage.bu, a.action, n, m, b, NP, lane);
        }
    };
    auto flush_tail = [&]() {   // slices kSteps (T / kSteps) .. T of the states, .. T - 1 of the actions
        if (ROLL_DIAG >= 1) return;
        if constexpr (kHalf) {
            tb_flush_half(s_stage, s_stage.bx, a.state, 0, n, T / kSteps, NP, lane);
            if (T % kSteps != 0) tb_flush_half(s_stage, s_stage.bu, a.action, n, m, T / kSteps, NP, lane);
        } else if constexpr (TB) {
            tb_flush(s_stage, s_stage.bx, a.state, 0, n, T / kTB, NP, lane);
            if (T % kTB != 0) tb_flush(s_stage, s_stage.bu, a.action, n, m, T / kTB, NP, lane);
        }
    };

    double x[N], xr[N], ur[M], kf[M], Kt[M * N];
    bool fin = true;
    double total = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < n) {
            x[i] = a.x0[gi<MINOR>(i, n, p, NP)];
            fin = fin && (x[i] < 1e8);
            if constexpr (TB) put_x(0, i, x[i]);
            else if constexpr (!CO) { if (FAST || a.state) st[gi<MINOR>(i, En1, p, NP)] = x[i]; }
        }
    if constexpr (FAST) {
        // The driver's variant.  Addresses are a wave-uniform slice pointer per array (advanced with scalar adds) plus
        // a constant 32-bit per-lane offset per element, and the inputs of step t+2 are loaded into the register set of
        // step t as soon as its control law has consumed them (two named sets; unconditionally: the last slice is simply
        // loaded again) - the generic loop below costs ~65 address / copy instructions per step, a sixth of the step.
        static_assert(MINOR && SYS::kStatic, "FAST is instantiated for the struct-of-arrays layout and static dimensions only");
        constexpr int kKq = KL ? M * N / 16 : 1;   // KL: 16-byte pieces of K_t a lane fetches for the wave
        struct In { double xr[N], ur[M], kf[M], K[KL ? 1 : M * N], w; kl_v2 kq[kKq]; };
        // Every load is  buffer_load_dwordx2 v, v_lane, s[descriptor of the array's current slice], s_element offen :
        // the descriptor's base is the wave-uniform slice pointer (advanced with two scalar adds a step), the element's
        // offset a loop-invariant SGPR and the lane's offset ONE VGPR per array - as a flat load every element cost a
        // 64-bit vector add (base + zero-extended lane offset) and a 64-bit VGPR pair to hold that offset.
        const unsigned np8 = (unsigned)(NP * 8), p8 = (unsigned)(p * 8);
        const unsigned vkf = KMINOR ? (unsigned)(bank * M) * np8 + p8 : (unsigned)((((size_t)bank * NP + p) * Em) * 8);
        const unsigned vK = KMINOR ? (unsigned)(bank * M * N) * np8 + p8 : (unsigned)((((size_t)bank * NP + p) * Enm) * 8);
        const size_t sx = (size_t)N * np8, su = (size_t)M * np8, skf = KMINOR ? kmul * su : (size_t)M * 8,
                     sK = KMINOR ? (size_t)kmul * N * M * np8 : (size_t)N * M * 8;
        const char *pxr = reinterpret_cast<const char*>(a.xref), *pur = reinterpret_cast<const char*>(a.uref),
                   *pK = reinterpret_cast<const char*>(a.K), *pkf = reinterpret_cast<const char*>(a.kff),
                   *pw = reinterpret_cast<const char*>(a.weighting);
        auto rs = [](const char* b) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b), 0, 0xffffffff, 0x00020000); };
        auto bl = [](__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so) {
            return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, v, so, 0));
        };
        double xtop = -INFINITY;
        // KL: the wave's idx range covers the problems p_lo .. p_lo + 7 at most (launch_rollout checks nA >= 8); lane l
        // fetches piece l % 8 of problem p_lo + l / 8 (clamped: a fetch without a reader), and reads slot p - p_lo
        unsigned vKq = 0, lds_put = 0, lds_get = 0;
        if constexpr (KL) {
            const size_t nA = (size_t)(a.nb_alphas - a.alpha_begin);
            const size_t p_lo = ((size_t)blockIdx.x * 64) / nA;
            const size_t pd = p_lo + lane / 8 < NP ? p_lo + lane / 8 : NP - 1;
            vKq = (unsigned)((((size_t)gains_bank(a.out_bank, pd) * NP + pd) * Enm) * 8) + (lane % 8) * (unsigned)(M * N);
            lds_put = (lane / 8) * (unsigned)(kKSlot / 2) + (lane % 8) * (unsigned)kKq;
            lds_get = (idle || p < p_lo || p - p_lo > 7 ? 0u : (unsigned)(p - p_lo)) * (unsigned)(kKSlot / 2);
        }
        int tl = 0;  // the slice the pointers address
        auto load = [&](In& s) {
            const __amdgpu_buffer_rsrc_t rx = rs(pxr), ru = rs(pur), rk = rs(pK), rf = rs(pkf);
#pragma unroll
            for (int i = 0; i < N; ++i) s.xr[i] = bl(rx, p8, (unsigned)i * np8);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                s.ur[j] = bl(ru, p8, (unsigned)j * np8);
                s.kf[j] = KMINOR ? bl(rf, vkf, (unsigned)j * np8) : bl(rf, vkf + 8u * j, 0u);
            }
            if constexpr (KL) {
#pragma unroll
                for (int k = 0; k < kKq; ++k)
                    s.kq[k] = __builtin_bit_cast(kl_v2, __builtin_amdgcn_raw_buffer_load_b128(rk, vKq + 16u * k, 0u, 0));
            } else {
#pragma unroll
                for (int k = 0; k < M * N; ++k) s.K[k] = KMINOR ? bl(rk, vK, (unsigned)k * np8) : bl(rk, vK + 8u * k, 0u);
            }
            s.w = *reinterpret_cast<const double*>(pw);
            if (tl + 1 < T) { pxr += sx; pur += su; pK += sK; pkf += skf; pw += 8; ++tl; }
        };
        double* sp[N];
        double* ap[M];
        if constexpr (!TB && !CO) {
#pragma unroll
            for (int i = 0; i < N; ++i) sp[i] = st + ((size_t)i * NP + p);
#pragma unroll
            for (int j = 0; j < M; ++j) ap[j] = ac + ((size_t)j * NP + p);
        }
        const size_t dsx = (size_t)N * NP, dsu = (size_t)M * NP;
        int t = 0;
        // put_next: the new state belongs to the block being filled (false only for the last step of a block, whose
        // new state the caller puts after the block has been written out)
        // The step's inputs are consumed by its first few instructions (the control law and the cost); the register set
        // is then refilled IN PLACE with the inputs of the step after next.  Two steps of distance matter because
        // loads and stores retire through one in-order counter: a load issued behind the write-out of a block is only
        // seen to have arrived once those 40 stores have been acknowledged.
        // (kAhead2: for the lone-wave forms.  Where several waves share a SIMD they hide that wait for each other, and
        // the plain one-step-ahead prefetch into the other register set needs fewer registers - an occupancy step for dx = 6.)
        constexpr bool kAhead2 = (!TB || (UNR && !kHalf)) && !CO;
        // (kOne: models with more than 8 rows a step - two register sets of their inputs (quadrotor: 68 doubles each)
        // do not fit next to the dynamics; ONE set, refilled in place with the next step's inputs as soon as the control
        // law has consumed it - they arrive while the dynamics integrate.)
        constexpr bool kOne = TB && (N + M > 8);
        auto step = [&](In& s, auto put_next_c) {
            constexpr bool put_next = decltype(put_next_c)::value;
            double u[M];  // u = clip(uref + alpha kff + K (x - xref))    objects.py:137-139, ilqr.py:97
            if constexpr (KL) {
                // the wave's LDS queue is in order: these writes follow the previous step's reads, the reads below follow them
#pragma unroll
                for (int k = 0; k < kKq; ++k) s_gains[lds_put + k] = s.kq[k];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double kx[M];
#pragma unroll
                for (int j = 0; j < M; ++j) kx[j] = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double dxi = x[i] - s.xr[i];
#pragma unroll
                    for (int j = 0; j < M; j += 2) {
                        const kl_v2 kk = s_gains[lds_get + (M * i + j) / 2];
                        kx[j] = fma(kk.x, dxi, kx[j]);
                        if (j + 1 < M) kx[j + 1] = fma(kk.y, dxi, kx[j + 1]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < M; ++j) u[j] = clip_hw(fma(alpha, s.kf[j], s.ur[j]) + kx[j], e.umax[j]);
            } else {
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    double kx = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) kx = fma(s.K[j + M * i], x[i] - s.xr[i], kx);
                    u[j] = clip_hw(fma(alpha, s.kf[j], s.ur[j]) + kx, e.umax[j]);
                }
            }
            total += env_cost_plain<N, M>(e, x, u, s.w);  // ilqr.py:98 passes u_last = u_t
            if constexpr (kAhead2 || kOne) load(s);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                if constexpr (TB) put_u(t, j, u[j]);
                else if constexpr (!CO) { *ap[j] = u[j]; ap[j] += dsu; }
            }
            double xn[N];
            SYS::dynamics_clipped_hw(e, x, u, xn);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                x[i] = xn[i];
                xtop = max_hw(xtop, x[i]);   // all(state < 1e8), ilqr.py:141: a clipped state is never NaN, so the largest one decides
                if constexpr (TB) { if constexpr (put_next) put_x(t + 1, i, x[i]); }
                else if constexpr (!CO) { sp[i] += dsx; *sp[i] = x[i]; }
            }
        };
        In sa, sb;
        load(sa);
        if constexpr (kAhead2) load(sb);
        if constexpr (TB) {
            // Whole blocks: 8 steps, then both blocks go out - unconditional code in a loop with one entry state.  As
            // many stores are put behind the first load as a block's write-out puts behind every later one (block 0 as
            // it stands, written again when it is complete): the waitcnt pass merges the loop's two entries, and with
            // no stores on the first it would make every block wait for ALL but the newest loads, i.e. for its own
            // write-out to be acknowledged.
            flush_block(0);
            const int nblk = T / kSteps;
            for (int b = 0; b < nblk; ++b) {
                if constexpr (kHalf) {
                    // four straight-line steps: a single loop whose only entries are the dummy write-out above and its
                    // own back edge, both with a half block's stores behind the loads of the next inputs
#pragma unroll
                    for (int k = 0; k < kHalfSteps / 2; ++k) {
                        if constexpr (!kAhead2 && !kOne) load(sb);
                        step(sa, std::true_type{});
                        ++t;
                        if constexpr (!kAhead2 && !kOne) load(sa);
                        if (k + 1 < kHalfSteps / 2) step(kOne ? sa : sb, std::true_type{});
                        else step(kOne ? sa : sb, std::false_type{});
                        ++t;
                    }
                } else if constexpr (UNR) {
#pragma unroll
                    for (int k = 0; k < kTB / 2; ++k) {
                        if constexpr (!kAhead2 && !kOne) load(sb);
                        step(sa, std::true_type{});
                        ++t;
                        if constexpr (!kAhead2 && !kOne) load(sa);
                        if (k + 1 < kTB / 2) step(sb, std::true_type{});
                        else step(sb, std::false_type{});
                        ++t;
                    }
                } else {
#pragma unroll 1
                    for (int k = 0; k < kTB / 2 - 1; ++k) {
                        if constexpr (!kAhead2 && !kOne) load(sb);
                        step(sa, std::true_type{});
                        ++t;
                        if constexpr (!kAhead2 && !kOne) load(sa);
                        step(kOne ? sa : sb, std::true_type{});
                        ++t;
                    }
                    if constexpr (!kAhead2 && !kOne) load(sb);
                    step(sa, std::true_type{});
                    ++t;
                    if constexpr (!kAhead2 && !kOne) load(sa);
                    step(kOne ? sa : sb, std::false_type{});
                    ++t;
                }
                flush_block(b);
#pragma unroll
                for (int i = 0; i < N; ++i) put_x(t, i, x[i]);   // slice 8 b + 8 = slot 0 of the next block
            }
            while (t < T) {   // the last, partial block: fewer than 8 (4) steps
                if constexpr (!kAhead2 && !kOne) load(sb);
                step(sa, std::true_type{});
                if (++t >= T) break;
                if constexpr (!kAhead2 && !kOne) load(sa);
                step(kOne ? sa : sb, std::true_type{});
                ++t;
            }
            flush_tail();
        } else {
            while (t < T) {
                if constexpr (!kAhead2 && !kOne) load(sb);
                step(sa, std::true_type{});
                if (++t >= T) break;
                if constexpr (!kAhead2 && !kOne) load(sa);
                step(sb, std::true_type{});
                ++t;
            }
        }
        double zero[M];
#pragma unroll
        for (int j = 0; j < M; ++j) zero[j] = 0.0;
        total += env_cost_plain<N, M>(e, x, zero, a.weighting[T]);  // ilqr.py:101
        fin = fin && (xtop < 1e8);
        if (live) {
            if (a.ret) a.ret[(size_t)ai * NP + p] = total;
            if (a.finite) a.finite[(size_t)ai * NP + p] = fin ? 1 : 0;
        }
        return;
    }
    auto load = [&](int t) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < n) xr[i] = a.xref[gi<MINOR>((size_t)n * t + i, En1, p, NP)];
#pragma unroll
        for (int j = 0; j < M; ++j)
            if (j < m) {
                ur[j] = a.uref[gi<MINOR>((size_t)m * t + j, Em, p, NP)];
                kf[j] = a.kff[gk<KMINOR>(m, T, t, j, p, NP, kmul, bank)];
            }
#pragma unroll
        for (int k = 0; k < M * N; ++k)
            if (k < n * m) Kt[k] = a.K[gk<KMINOR>((size_t)n * m, T, t, k, p, NP, kmul, bank)];
    };
    if (T > 0) load(0);
    double w = a.weighting[0];
    for (int t = 0; t < T; ++t) {
        // u = clip(uref + alpha kff + K (x - xref))    objects.py:137-139, ilqr.py:97
        double u[M];
#pragma unroll
        for (int j = 0; j < M; ++j)
            if (j < m) {
                double s = fma(alpha, kf[j], ur[j]);
                double kx = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (i < n) kx = fma(Kt[j + m * i], x[i] - xr[i], kx);
                u[j] = clip_g(s + kx, e.umax[j]);
            }
        const double wt = w;
        if (t + 1 < T) load(t + 1);  // prefetch the next step's gains while this one integrates
        w = a.weighting[t + 1];
        const double c = env_cost<N, M>(e, x, u, u, wt, n, m);  // ilqr.py:98 passes u_last = u_t
        total += c;
        if (a.cost) co[gi<MINOR>(t, T + 1, p, NP)] = c;
        if constexpr (TB) {
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j < m) put_u(t, j, u[j]);
        } else if (a.action) {
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j < m) ac[gi<MINOR>((size_t)m * t + j, Em, p, NP)] = u[j];
        }
        double xn[N];
        SYS::dynamics_clipped(e, x, u, xn);  // u is clipped above
        if constexpr (TB) {
            if ((t & (kSteps - 1)) == kSteps - 1) flush_block(t / kSteps);
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < n) {
                x[i] = xn[i];
                fin = fin && (x[i] < 1e8);
                if constexpr (TB) put_x(t + 1, i, x[i]);
                else if (a.state) st[gi<MINOR>((size_t)n * (t + 1) + i, En1, p, NP)] = x[i];
            }
    }
    if constexpr (TB) flush_tail();
    {
        double zero[M];
#pragma unroll
        for (int j = 0; j < M; ++j) zero[j] = 0.0;
        const double c = env_cost<N, M>(e, x, zero, zero, w, n, m);  // ilqr.py:101
        total += c;
        if (a.cost) co[gi<MINOR>(T, T + 1, p, NP)] = c;
    }
    if (live) {
        if (a.ret) a.ret[(size_t)ai * NP + p] = total;
        if (a.finite) a.finite[(size_t)ai * NP + p] = fin ? 1 : 0;
    }
}

struct StepArgs {
    trajopt_env_t env;
    const double *x, *u, *u_last, *a;
    double *xn, *cost;
    size_t NP;
};

template <class SYS, bool MINOR>
__global__ void env_step_kernel(StepArgs s) {
    constexpr int N = SYS::N, M = SYS::M;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= s.NP) return;
    const trajopt_env_t& e = s.env;
    const int n = SYS::n(e), m = SYS::m(e);
    double x[N], u[M], ul[M], xn[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = (i < n) ? s.x[gi<MINOR>(i, n, p, s.NP)] : 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        u[j] = (j < m) ? s.u[gi<MINOR>(j, m, p, s.NP)] : 0.0;
        ul[j] = (j < m && s.u_last) ? s.u_last[gi<MINOR>(j, m, p, s.NP)] : 0.0;
    }
    if (s.cost) s.cost[p] = env_cost<N, M>(e, x, u, ul, s.a ? s.a[p] : 1.0, n, m);
    if (s.xn) {
        SYS::dynamics(e, x, u, xn);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < n) s.xn[gi<MINOR>(i, n, p, s.NP)] = xn[i];
    }
}

// The lone-wave form of the driver's rollout (UNR) is taken while every wave has a SIMD to itself.
bool rollout_straight_line(int n, int m, size_t NP, int nb_alphas) {
    return (double)NP * nb_alphas <= 64.0 * 1024.0 && n + m <= 7;
}

// KL form of the driver's rollout (gains through LDS): see rollout_kernel.  alpha_minor implies PROBLEM_MAJOR gains,
// no problem list and no second pass; with >= 8 alphas 64 consecutive (problem, alpha) pairs hold <= 8 problems.
#ifndef ROLL_NO_KL
#define ROLL_NO_KL 0
#endif
static bool rollout_gains_through_lds(int n, int m, int alpha_minor, int nb_alphas) {
    return !ROLL_NO_KL && alpha_minor && nb_alphas >= 8 && n + m > 8 && (n * m) % 16 == 0;
}

// The returns-only form (CO) exists for the variant the driver's plain-cost call takes (FAST in rollout_kernel).
bool rollout_cost_only_supported(const trajopt_env_t* env, int T, int64_t NP, int k_layout) {
    if (env->model == TRAJOPT_ENV_LQR || env->slew_rate || env->periodic != 0 || env->cost_times_dt || env->nb_features != 0 || T <= 0)
        return false;
    const double nm = (double)env->dm_state * env->dm_act;
    const double kbytes = (double)TRAJOPT_GAIN_BANKS * (k_layout == TRAJOPT_LAYOUT_PROBLEM_MINOR ? nm * (double)NP * 8.0 : (double)NP * nm * (double)T * 8.0);
    return kbytes + (double)NP * 8.0 < 4294967296.0;
}

struct RolloutLaunch {
    RolloutArgs a;
    bool tb;   // time-blocked candidate planes (the driver's call)
    bool cost_only;   // no trajectories (CO)
    int err = TRAJOPT_OK;
    int layout, k_layout;
    hipStream_t st;
    template <class SYS>
    void operator()() {
        const size_t nA = a.sel ? 1 : (size_t)(a.nb_alphas - a.alpha_begin);
        const size_t total = a.NP * nA;
        const size_t tiles8 = (a.NP + 511) / 512;   // tiles of 64 problems, rounded up to a multiple of 8 (one per XCD)
        const dim3 grid((unsigned)(a.xcd_tiles ? tiles8 * 8 * nA : (total + 63) / 64)), block(64);
        const bool mi = layout == TRAJOPT_LAYOUT_PROBLEM_MINOR, kmi = k_layout == TRAJOPT_LAYOUT_PROBLEM_MINOR;
        // (FAST addresses the gains with 32-bit per-lane byte offsets)
        const double kbytes = (a.out_bank ? (double)TRAJOPT_GAIN_BANKS : 1.0) * (kmi ? (double)SYS::N * SYS::M * (double)a.NP * 8.0
                                                                                     : (double)a.NP * SYS::N * SYS::M * (double)a.T * 8.0);
        const bool fast = SYS::kStatic && (cost_only || (a.state && a.action)) && !a.cost && !a.env.slew_rate && a.env.periodic == 0 &&
                          !a.env.cost_times_dt && a.env.nb_features == 0 && a.T > 0 && kbytes + (double)a.NP * 8.0 < 4294967296.0;
        if constexpr (SYS::kStatic) {
            if (mi && fast && cost_only) {
                if (kmi) hipLaunchKernelGGL((rollout_kernel<SYS, true, true, true, false, false, true>), grid, block, 0, st, a);
                else hipLaunchKernelGGL((rollout_kernel<SYS, true, false, true, false, false, true>), grid, block, 0, st, a);
                return;
            }
            if (mi && fast) {
                // straight-line blocks while every wave has a SIMD to itself (and the model's registers allow it)
                // (tried for dx + du >= 6, where the stage's LDS allows one wave per SIMD anyway: 8192 double cartpoles
                // 1.90 ms against 1.81 rolled)
                const bool unr = rollout_straight_line(SYS::N, SYS::M, a.NP, (int)nA);
                if (tb && unr && kmi) hipLaunchKernelGGL((rollout_kernel<SYS, true, true, true, true, true>), grid, block, 0, st, a);
                else if (tb && unr) hipLaunchKernelGGL((rollout_kernel<SYS, true, false, true, true, true>), grid, block, 0, st, a);
                else if (tb && kmi) hipLaunchKernelGGL((rollout_kernel<SYS, true, true, true, true, false>), grid, block, 0, st, a);
                else if (tb && rollout_gains_through_lds(SYS::N, SYS::M, a.alpha_minor, (int)nA)) {
                    if constexpr (SYS::N + SYS::M > 8 && (SYS::N * SYS::M) % 16 == 0)
                        hipLaunchKernelGGL((rollout_kernel<SYS, true, false, true, true, false, false, true>), grid, block, 0, st, a);
                }
                else if (tb) hipLaunchKernelGGL((rollout_kernel<SYS, true, false, true, true, false>), grid, block, 0, st, a);
                else if (kmi) hipLaunchKernelGGL((rollout_kernel<SYS, true, true, true, false, false>), grid, block, 0, st, a);
                else hipLaunchKernelGGL((rollout_kernel<SYS, true, false, true, false, false>), grid, block, 0, st, a);
                return;
            }
        }
        if (cost_only) { err = TRAJOPT_ERR_UNSUPPORTED; return; }   // (no returns-only form of the general kernel)
        if (tb && kmi) hipLaunchKernelGGL((rollout_kernel<SYS, true, true, false, true, false>), grid, block, 0, st, a);   // (tb: mi)
        else if (tb) hipLaunchKernelGGL((rollout_kernel<SYS, true, false, false, true, false>), grid, block, 0, st, a);
        else if (mi && kmi) hipLaunchKernelGGL((rollout_kernel<SYS, true, true, false, false, false>), grid, block, 0, st, a);
        else if (mi) hipLaunchKernelGGL((rollout_kernel<SYS, true, false, false, false, false>), grid, block, 0, st, a);
        else if (kmi) hipLaunchKernelGGL((rollout_kernel<SYS, false, true, false, false, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((rollout_kernel<SYS, false, false, false, false, false>), grid, block, 0, st, a);
    }
};

// Closed-loop episodes under an absolute time-varying law (row A9's forward pass):
//   Riccati.forward_pass / .rollout   /root/reference/trajopt/riccati/riccati.py:52-98
//   LinearControl.action              /root/reference/trajopt/riccati/objects.py:128-129   u_t = kff_t + K_t x_t
//   env.step                          e.g. /root/reference/trajopt/envs/pendulum/pendulum.py:131-138
// One thread per episode: cost_t = env.cost(x_t, u_t, u_{t-1}, w_t) on the UNCLIPPED action (u_{-1} = 0: the
// reference reads the still-zero last column through Python's negative index), x_{t+1} = env.dynamics(x_t, u_t) + noise_t
// (dynamics clips the action itself), cost_T = env.cost(x_T, 0, 0, w_T).  The noise is drawn by the caller (the env's
// own generator, so that the episode is the one the reference's loop produces) and added here.
// K (m, n, T) and kff (m, T) are ONE controller in the reference layout, shared by every episode; x0 (n, N), noise
// (T, n, N), state (T + 1, n, N), action (T, m, N), xnext (T, n, N; may be NULL), cost (T + 1, N) are PROBLEM_MINOR.
struct EpisodeArgs {
    trajopt_env_t env;
    const double *x0, *K, *kff, *noise, *weighting;
    double *state, *action, *xnext, *cost;
    size_t NP;
    int T;
};

template <class SYS>
__global__ __launch_bounds__(64) void episode_kernel(EpisodeArgs a) {
    constexpr int N = SYS::N, M = SYS::M;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x, NP = a.NP;
    if (p >= NP) return;
    const trajopt_env_t& e = a.env;
    const int n = SYS::n(e), m = SYS::m(e), T = a.T;
    double x[N], u[M], ul[M], xn[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = i < n ? a.x0[(size_t)i * NP + p] : 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) ul[j] = 0.0;
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double s = 0.0;
            if (j < m) {
                s = a.kff[(size_t)m * t + j];
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (i < n) s = fma(a.K[(size_t)m * n * t + j + (size_t)m * i], x[i], s);
            }
            u[j] = s;
        }
        a.cost[(size_t)t * NP + p] = env_cost<N, M>(e, x, u, ul, a.weighting[t], n, m);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < n) a.state[((size_t)n * t + i) * NP + p] = x[i];
#pragma unroll
        for (int j = 0; j < M; ++j)
            if (j < m) a.action[((size_t)m * t + j) * NP + p] = u[j];
        SYS::dynamics(e, x, u, xn);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < n) {
                x[i] = xn[i] + (a.noise ? a.noise[((size_t)n * t + i) * NP + p] : 0.0);
                if (a.xnext) a.xnext[((size_t)n * t + i) * NP + p] = x[i];
            }
#pragma unroll
        for (int j = 0; j < M; ++j) ul[j] = u[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < n) a.state[((size_t)n * T + i) * NP + p] = x[i];
    double zero[M];
#pragma unroll
    for (int j = 0; j < M; ++j) zero[j] = 0.0;
    a.cost[(size_t)T * NP + p] = env_cost<N, M>(e, x, zero, zero, a.weighting[T], n, m);
}

struct EpisodeLaunch {
    EpisodeArgs a;
    hipStream_t st;
    template <class SYS>
    void operator()() {
        hipLaunchKernelGGL((episode_kernel<SYS>), dim3((unsigned)((a.NP + 63) / 64)), dim3(64), 0, st, a);
    }
};

struct StepLaunch {
    StepArgs a;
    int layout;
    hipStream_t st;
    template <class SYS>
    void operator()() {
        const dim3 grid((unsigned)((a.NP + 63) / 64)), block(64);
        if (layout == TRAJOPT_LAYOUT_PROBLEM_MINOR) hipLaunchKernelGGL((env_step_kernel<SYS, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((env_step_kernel<SYS, false>), grid, block, 0, st, a);
    }
};

int launch_rollout(const trajopt_env_t* env, int T, int64_t NP, int layout, const double* x0, const double* K,
                   const double* kff, const double* xref, const double* uref, const double* weighting,
                   const double* alphas_host, int nb_alphas, const int32_t* active, double* state, double* action,
                   double* cost, double* ret, int32_t* finite, hipStream_t st, int k_layout, int alpha_begin,
                   int alpha_end, const int32_t* plist, const int32_t* pcount, const RolloutPlanes* planes) {
    if (alpha_begin < 0 || alpha_end > nb_alphas || alpha_begin >= alpha_end) return TRAJOPT_ERR_BAD_ARG;
    if (planes && (layout != TRAJOPT_LAYOUT_PROBLEM_MINOR || !state || !action)) return TRAJOPT_ERR_BAD_ARG;
    RolloutLaunch L;
    L.a.alpha_begin = alpha_begin; L.a.plist = plist; L.a.pcount = pcount;
    L.a.first_plane = planes ? planes->first_plane : 0; L.a.junk_plane = planes ? planes->junk_plane : 0;
    L.a.out_bank = planes ? planes->out_bank : nullptr;
    // (only where the chip is oversubscribed: measured on 4096 cartpoles x 11 alphas, 704 waves, the plain order is 5 % faster)
    L.a.sel = planes ? planes->sel : nullptr;
    L.cost_only = planes && planes->cost_only;
    if (L.cost_only && (L.a.sel || !rollout_cost_only_supported(env, T, NP, k_layout))) return TRAJOPT_ERR_UNSUPPORTED;
    L.a.xcd_tiles = (planes && !L.a.sel && (double)NP * (alpha_end - alpha_begin) > 65536.0) ? 1 : 0;
    L.tb = planes != nullptr && !L.cost_only;
    L.a.alpha_minor = (planes && !L.a.sel && !plist && !L.a.xcd_tiles && layout == TRAJOPT_LAYOUT_PROBLEM_MINOR &&
                       k_layout == TRAJOPT_LAYOUT_PROBLEM_MAJOR) ? 1 : 0;
    nb_alphas = alpha_end;  // the kernel runs alphas [alpha_begin, alpha_end)
    L.a.env = *env;
    L.a.x0 = x0; L.a.K = K; L.a.kff = kff; L.a.xref = xref; L.a.uref = uref; L.a.weighting = weighting;
    L.a.active = active; L.a.state = state; L.a.action = action; L.a.cost = cost; L.a.ret = ret; L.a.finite = finite;
    for (int i = 0; i < TRAJOPT_MAX_ALPHAS; ++i) L.a.alphas[i] = i < nb_alphas ? alphas_host[i] : 0.0;
    L.a.NP = (size_t)NP; L.a.T = T; L.a.nb_alphas = nb_alphas;
    L.layout = layout; L.k_layout = k_layout; L.st = st;
    if (!dispatch_model(*env, L)) return TRAJOPT_ERR_UNSUPPORTED;
    if (L.err != TRAJOPT_OK) return L.err;
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

}  // namespace trajopt

using namespace trajopt;

extern "C" {

int trajopt_ilqr_forward_pass_batched(const trajopt_env_t* env, int nb_steps, int64_t nb_problems, int layout,
                                      const double* x0, const double* K, const double* kff, const double* xref,
                                      const double* uref, const double* weighting, const double* alphas_host,
                                      int nb_alphas, const int32_t* active, double* state, double* action,
                                      double* cost, double* ret, int32_t* finite, void* stream) {
    if (!env || !x0 || !K || !kff || !xref || !uref || !weighting || !alphas_host) return TRAJOPT_ERR_BAD_ARG;
    if (nb_steps <= 0 || nb_problems <= 0 || nb_alphas <= 0 || nb_alphas > TRAJOPT_MAX_ALPHAS) return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
    return launch_rollout(env, nb_steps, nb_problems, layout, x0, K, kff, xref, uref, weighting, alphas_host, nb_alphas,
                          active, state, action, cost, ret, finite, as_stream(stream), layout, 0, nb_alphas, nullptr,
                          nullptr, nullptr);
}

int trajopt_riccati_forward_pass_batched(const trajopt_env_t* env, int nb_steps, int64_t nb_episodes, const double* x0,
                                         const double* K, const double* kff, const double* noise,
                                         const double* weighting, double* state, double* action, double* xnext,
                                         double* cost, void* stream) {
    if (!env || !x0 || !K || !kff || !weighting || !state || !action || !cost) return TRAJOPT_ERR_BAD_ARG;
    if (nb_steps <= 0 || nb_episodes <= 0) return TRAJOPT_ERR_BAD_ARG;
    EpisodeLaunch L;
    L.a.env = *env; L.a.x0 = x0; L.a.K = K; L.a.kff = kff; L.a.noise = noise; L.a.weighting = weighting;
    L.a.state = state; L.a.action = action; L.a.xnext = xnext; L.a.cost = cost;
    L.a.NP = (size_t)nb_episodes; L.a.T = nb_steps; L.st = as_stream(stream);
    if (!dispatch_model(*env, L)) return TRAJOPT_ERR_UNSUPPORTED;
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

int trajopt_env_step_batched(const trajopt_env_t* env, const double* x, const double* u, const double* u_last,
                             const double* a, int64_t nb_problems, int layout, double* xn, double* cost, void* stream) {
    if (!env || !x || !u || nb_problems <= 0) return TRAJOPT_ERR_BAD_ARG;
    if (layout != TRAJOPT_LAYOUT_PROBLEM_MAJOR && layout != TRAJOPT_LAYOUT_PROBLEM_MINOR) return TRAJOPT_ERR_BAD_ARG;
    StepLaunch L;
    L.a.env = *env; L.a.x = x; L.a.u = u; L.a.u_last = u_last; L.a.a = a; L.a.xn = xn; L.a.cost = cost;
    L.a.NP = (size_t)nb_problems; L.layout = layout; L.st = as_stream(stream);
    if (!dispatch_model(*env, L)) return TRAJOPT_ERR_UNSUPPORTED;
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

}  // extern "C"
