// Column-per-lane regularised Riccati backward pass: eight lanes per problem, eight problems per wave, for dx = 6,
// du = 1 (BASELINE config C3: double cartpoles) at batch sizes that leave the lane-per-problem kernel (bwd_tpp.hip)
// with a fraction of the SIMDs.
//
// Row A1 of the hot path: the recursion of /root/reference/trajopt/ilqr/src/util.cpp:123-162, with the lambda retry
// loop of /root/reference/trajopt/ilqr/ilqr.py:162-173 optionally folded in.
//
// Why: the recursion is a T-step dependency chain.  With lane = problem a step of a dx = 6 problem is ~1000 dependent
// fp64 instructions of ONE wave (5.3 cycles each on a SIMD's vector pipe whatever the number of active lanes), and 8192
// problems are 128 waves on 1024 SIMDs: 1.3 ms at C3, none of it bandwidth.  Here lane (q, j) of a wave owns column j
// of problem q's Hessian blocks:
//     P_j  = V A_j                          Qxx_:j = Cxx_:j + A' P_j         qx_j = cx_j + A_j' v
//     Z    = V B,  Quu, qu, Qux = Cxu' + Z' A,  K, kff          (every lane of the problem, redundantly: these are
//                                                                 cheaper to recompute than to pass between lanes)
//     W_:j = Qxx_:j + K' (Quu K_j + Qux_j) + Qux' K_j           v_j = qx_j + K_j (Quu kff + qu) + Qux_j kff
// and one LDS exchange per step (the wave's own 5 KB, no barrier: a wave's LDS queue is in order) hands every lane the
// whole of W, v and the next step's A; V = (W + W') / 2 is then formed in registers by every lane (util.cpp:161 - kept,
// the inputs need not be symmetric).  A step is ~330 instructions of a wave that carries 8 problems, and 8192 problems
// are 1024 waves - one per SIMD.  Beyond ~24k problems the redundancy costs more than the parallelism gains and the
// dispatcher (api.hip) goes back to lane = problem.
//
// PROBLEM_MINOR layout only (element e of step t for consecutive problems is contiguous: the 8 problems of a wave read
// 64-byte segments).  Inputs are fetched two steps ahead of the arithmetic, unconditionally and with clamped indices;
// stores go through range-checked buffer descriptors and a lane with nothing to store passes an offset beyond the
// range (unconditional instructions: see fused_qb4.hip on hipcc's wait-count pass).
#include "smallmat.h"

namespace trajopt {

struct Col8Args {
    trajopt_backward_io_t io;
    trajopt_lambda_schedule_t sched;
    size_t NP;
    int T, reg, retry;
};

typedef double c8_d2 __attribute__((ext_vector_type(2)));
typedef unsigned c8_v2u __attribute__((ext_vector_type(2)));
constexpr unsigned kC8NoStore = 0xfffffff8u;  // beyond any array this kernel takes (the launcher checks the sizes)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t c8_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double c8_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void c8_st(double x, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(c8_v2u, x), r, off, 0, 0);
}

// What a lane fetches for one step: its own column of A (and of Cxx), its own element of B, cx (and Cxu), the
// problem's scalars.  B (and Cxu) reach the other lanes of the problem through the LDS exchange, like A.
template <int N>
struct Col8In {
    double a[N], cxxo[N], bo, cxuo, cuu, cxo, cu, ur, w;
};

template <int N, bool FULL, bool GN, bool SC>
__global__ __launch_bounds__(64) void bwd_col8_kernel(Col8Args a) {
    constexpr int NN = N * N;
    // exchange buffer of one problem: W (N x N, column-major) | v (N) | of the next step: A (N x N) | B (N) | Cxu (N)
    constexpr int oV_ = NN, oA_ = NN + N, oB_ = 2 * NN + N, oX_ = 2 * NN + 2 * N, PS = 2 * NN + 3 * N;
    static_assert(N == 6, "bank-conflict-free problem stride and 16-byte column alignment are worked out for dx = 6");
    static_assert(N % 2 == 0 && PS % 2 == 0, "16-byte LDS accesses");
    // PS = 90 doubles = 180 dwords = 52 (mod 64 banks): the eight problems' 16-byte broadcast reads hit disjoint banks
    // (+ one problem-sized dump per lane 6 / 7 of a problem: their LDS writes are the same unconditional instructions)
    __shared__ __align__(16) double xch[8 * PS + 16 * PS];

    const unsigned lane = threadIdx.x, q = lane >> 3, j = lane & 7u;
    const size_t NP = a.NP;
    const int T = a.T;
    const trajopt_backward_io_t& io = a.io;
    const size_t p0 = (size_t)blockIdx.x * 8;
    const bool exists = p0 + q < NP;
    const unsigned qc = exists ? q : (unsigned)(NP - 1 - p0);  // padding lanes recompute the last problem, store nothing
    const size_t p = p0 + qc;
    const bool skip = !exists || (io.active && io.active[p] == 0);
    const bool col_lane = j < (unsigned)N, s_lane = j == (unsigned)N;
    const unsigned jc = col_lane ? j : (unsigned)(N - 1);  // lanes 6, 7 of a problem run along on column 5
    double* xq = xch + q * PS;
    c8_d2* xq2 = reinterpret_cast<c8_d2*>(xq);
    // where this lane writes its column: column j of the problem's buffer, or column 0 of its own dump
    double* wq = col_lane ? xq : xch + (8 + 2 * q + (j - N)) * PS;
    c8_d2* wq2 = reinterpret_cast<c8_d2*>(wq);
    const unsigned jw = col_lane ? j : 0u;

    const unsigned np8 = (unsigned)(NP * 8);
    const unsigned vo_q = qc * 8u;                           // element 0 of the problem
    const unsigned vo_col = (unsigned)N * jc * np8 + vo_q;   // element (0, jc) of an N x N array
    const unsigned vo_own = jc * np8 + vo_q;                 // element jc of an N-vector
    const unsigned p8 = (unsigned)(p * 8);

    // gains: bank io.out_bank[p] of those interleaved per time step (trajopt_backward_io_t::out_bank)
    const unsigned kmul = io.out_bank ? (unsigned)TRAJOPT_GAIN_BANKS : 1u;
    const unsigned bank = exists ? (unsigned)gains_bank(io.out_bank, p) : 0u;
    const unsigned oK0 = (bank * N + jc) * np8 + p8, okff0 = bank * np8 + p8;
    const unsigned sK = kmul * (unsigned)N * np8, skff = kmul * np8;
    const __amdgpu_buffer_rsrc_t rK = c8_rsrc(io.K, (unsigned)((size_t)kmul * N * T * NP * 8));
    const __amdgpu_buffer_rsrc_t rkff = c8_rsrc(io.kff, (unsigned)((size_t)kmul * T * NP * 8));
    // full outputs (FULL only; a null array gets an empty range: every store to it is dropped)
    const __amdgpu_buffer_rsrc_t rQxx = c8_rsrc(io.Qxx ? io.Qxx : io.K, io.Qxx ? (unsigned)((size_t)NN * T * NP * 8) : 0u);
    const __amdgpu_buffer_rsrc_t rQux = c8_rsrc(io.Qux ? io.Qux : io.K, io.Qux ? (unsigned)((size_t)N * T * NP * 8) : 0u);
    const __amdgpu_buffer_rsrc_t rQuu = c8_rsrc(io.Quu ? io.Quu : io.K, io.Quu ? (unsigned)((size_t)T * NP * 8) : 0u);
    const __amdgpu_buffer_rsrc_t rqx = c8_rsrc(io.qx ? io.qx : io.K, io.qx ? (unsigned)((size_t)N * T * NP * 8) : 0u);
    const __amdgpu_buffer_rsrc_t rqu = c8_rsrc(io.qu ? io.qu : io.K, io.qu ? (unsigned)((size_t)T * NP * 8) : 0u);
    const __amdgpu_buffer_rsrc_t rV = c8_rsrc(io.V ? io.V : io.K, io.V ? (unsigned)((size_t)NN * (T + 1) * NP * 8) : 0u);
    const __amdgpu_buffer_rsrc_t rv = c8_rsrc(io.v ? io.v : io.K, io.v ? (unsigned)((size_t)N * (T + 1) * NP * 8) : 0u);

    double gsel[N];  // structured cost: column jc of diag(gxx)
    double guu = 0.0;
    if constexpr (SC) {
#pragma unroll
        for (int i = 0; i < N; ++i) gsel[i] = ((unsigned)i == jc) ? io.cost_gxx[i] : 0.0;
        guu = io.cost_guu[0];
    }

    // Inputs: one descriptor per array for the whole kernel (base = the wave's first problem), the time step in the
    // scalar offset, element and lane in the vector offset (every array is below 4 GB: col8_shape_ok).
    const void* nul = io.K;
    const __amdgpu_buffer_rsrc_t rA = c8_rsrc(io.A + p0, 0xffffffffu), rB = c8_rsrc(io.B + p0, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t rcx = c8_rsrc(io.cx + p0, 0xffffffffu), rcu = c8_rsrc(io.cu + p0, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t rur = c8_rsrc(GN ? (const void*)(io.uref + p0) : nul, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t rCxx = c8_rsrc(SC ? nul : (const void*)(io.Cxx + p0), 0xffffffffu);
    const __amdgpu_buffer_rsrc_t rCxu = c8_rsrc(SC ? nul : (const void*)(io.Cxu + p0), 0xffffffffu);
    const __amdgpu_buffer_rsrc_t rCuu = c8_rsrc(SC ? nul : (const void*)(io.Cuu + p0), 0xffffffffu);
    unsigned vo_ci[N];
#pragma unroll
    for (int i = 0; i < N; ++i) vo_ci[i] = vo_col + (unsigned)i * np8;
    auto fetch = [&](Col8In<N>& s, int t) {
        // (t is wave-uniform; saying so keeps the scalar offsets in SGPRs - hipcc's analysis gives up on the sweep loop)
        const unsigned tu = (unsigned)__builtin_amdgcn_readfirstlane(t);
        const unsigned s1 = tu * np8, sn = s1 * (unsigned)N, snn = sn * (unsigned)N;
#pragma unroll
        for (int i = 0; i < N; ++i) s.a[i] = c8_ld(rA, vo_ci[i], snn);
        s.bo = c8_ld(rB, vo_own, sn);
        if constexpr (SC) {
            s.w = io.cost_w[tu];  // wave-uniform
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) s.cxxo[i] = c8_ld(rCxx, vo_ci[i], snn);
            s.cxuo = c8_ld(rCxu, vo_own, sn);
            s.cuu = c8_ld(rCuu, vo_q, s1);
        }
        s.cxo = c8_ld(rcx, vo_own, sn);
        s.cu = c8_ld(rcu, vo_q, s1);
        if constexpr (GN) s.ur = c8_ld(rur, vo_q, s1);
    };

    double lam = io.lmbda[p];
    double dlam = (a.retry && io.dlmbda) ? io.dlmbda[p] : 1.0;
    int div = -1, retries = 0;
    double dV0 = 0.0, dV1 = 0.0, gmax = -INFINITY;
    bool frozen = skip;

    for (;;) {  // sweeps: one, plus one per lambda retry (ilqr.py:162-173)
        bool alive = !frozen;
        int sdiv = -1;
        double acc0 = 0.0, acc1 = 0.0, gm = -INFINITY;
        double V[NN], v[N], Bv[N], Xv[N];   // Xv: Cxu, dense cost only
        Col8In<N> s0, s1, s2;
        auto put_inputs = [&](const Col8In<N>& s) {
#pragma unroll
            for (int i = 0; i < N / 2; ++i) wq2[(oA_ + N * jw) / 2 + i] = c8_d2{s.a[2 * i], s.a[2 * i + 1]};
            wq[oB_ + jw] = s.bo;
            if constexpr (!SC) wq[oX_ + jw] = s.cxuo;
        };
        // A of the current step stays in LDS and is read a column at a time where it is used (36 doubles held in
        // registers across the step cost more in AGPR round trips than the 18 reads); B (and Cxu) are read back whole
        auto get_col = [&](double (&ac)[N], int i) {
#pragma unroll
            for (int h = 0; h < N / 2; ++h) {
                const c8_d2 x = xq2[(oA_ + N * i) / 2 + h];
                ac[2 * h] = x[0];
                ac[2 * h + 1] = x[1];
            }
        };
        auto get_inputs = [&](double (&B_)[N], double (&X_)[N]) {
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                const c8_d2 x = xq2[oB_ / 2 + i];
                B_[2 * i] = x[0];
                B_[2 * i + 1] = x[1];
                if constexpr (!SC) {
                    const c8_d2 y = xq2[oX_ / 2 + i];
                    X_[2 * i] = y[0];
                    X_[2 * i + 1] = y[1];
                }
            }
        };
        fetch(s0, T - 1);
        fetch(s1, T >= 2 ? T - 2 : 0);
        {
            // terminal condition (util.cpp:123-124)
            const __amdgpu_buffer_rsrc_t rcT = c8_rsrc(io.cx + (size_t)T * N * NP + p0, 0xffffffffu);
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = c8_ld(rcT, vo_q, (unsigned)i * np8);
            double Vo[N];
            if constexpr (SC) {
                const double w = io.cost_w[T];
#pragma unroll
                for (int k = 0; k < N; ++k)
#pragma unroll
                    for (int i = 0; i < N; ++i) V[i + N * k] = (i == k) ? w * io.cost_gxx[i] : 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) Vo[i] = w * gsel[i];
            } else {
                const __amdgpu_buffer_rsrc_t rCT = c8_rsrc(io.Cxx + (size_t)T * NN * NP + p0, 0xffffffffu);
#pragma unroll
                for (int e = 0; e < NN; ++e) V[e] = c8_ld(rCT, vo_q, (unsigned)e * np8);
#pragma unroll
                for (int i = 0; i < N; ++i) Vo[i] = c8_ld(rCT, vo_col, (unsigned)i * np8);
            }
            if constexpr (FULL) {
                const bool st = alive && col_lane;
                const unsigned oV = (unsigned)(((size_t)T * NN + N * jc) * NP * 8) + p8;
#pragma unroll
                for (int i = 0; i < N; ++i) c8_st(Vo[i], rV, st ? oV + (unsigned)i * np8 : kC8NoStore);
                c8_st(c8_ld(rcT, vo_own, 0), rv, st ? (unsigned)(((size_t)T * N + jc) * NP * 8) + p8 : kC8NoStore);
            }
            // A, B (Cxu) of step T-1 to every lane of the problem
            put_inputs(s0);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            get_inputs(Bv, Xv);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        unsigned oK = oK0 + (unsigned)(T - 1) * sK, okff = okff0 + (unsigned)(T - 1) * skff;

        // one step of util.cpp:128-161 with the inputs `s`; `nx` holds the (landed) inputs of the step after it
        auto step = [&](const Col8In<N>& s, const Col8In<N>& nx, int t) {
            double Z[N], P[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double z = V[i] * Bv[0], pp = V[i] * s.a[0];   // (a product, not fma(.., 0.0): no zero to materialise)
#pragma unroll
                for (int k = 1; k < N; ++k) {
                    z = fma(V[i + N * k], Bv[k], z);
                    pp = fma(V[i + N * k], s.a[k], pp);
                }
                Z[i] = z;
                P[i] = pp;
            }
            double Quu = SC ? guu : s.cuu, qu = s.cu;   // util.cpp:129,133
#pragma unroll
            for (int k = 0; k < N; ++k) {
                Quu = fma(Bv[k], Z[k], Quu);
                qu = fma(Bv[k], v[k], qu);
            }
            double Qux[N], Qxxo[N];                      // util.cpp:130 (all of it), util.cpp:128 (own column)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double ac[N];
                get_col(ac, i);
                double x = SC ? ac[0] * Z[0] : fma(ac[0], Z[0], Xv[i]);
                double y = fma(ac[0], P[0], SC ? s.w * gsel[i] : s.cxxo[i]);
#pragma unroll
                for (int k = 1; k < N; ++k) {
                    x = fma(ac[k], Z[k], x);
                    y = fma(ac[k], P[k], y);
                }
                Qux[i] = x;
                Qxxo[i] = y;
            }
            double Quxo = SC ? s.a[0] * Z[0] : fma(s.a[0], Z[0], s.cxuo), qxo = fma(s.a[0], v[0], s.cxo);  // own elements of Qux and qx (util.cpp:132)
#pragma unroll
            for (int k = 1; k < N; ++k) {
                Quxo = fma(s.a[k], Z[k], Quxo);
                qxo = fma(s.a[k], v[k], qxo);
            }
            if constexpr (FULL) {
                // Q-function blocks are stored before the PD test, like the reference (util.cpp:128-133)
                const bool stc = alive && col_lane, sts = alive && s_lane;
                const unsigned oQ = (unsigned)(((size_t)t * NN + N * jc) * NP * 8) + p8;
#pragma unroll
                for (int i = 0; i < N; ++i) c8_st(Qxxo[i], rQxx, stc ? oQ + (unsigned)i * np8 : kC8NoStore);
                const unsigned ov = (unsigned)(((size_t)t * N + jc) * NP * 8) + p8, o1 = (unsigned)((size_t)t * NP * 8) + p8;
                c8_st(Quxo, rQux, stc ? ov : kC8NoStore);
                c8_st(qxo, rqx, stc ? ov : kC8NoStore);
                c8_st(Quu, rQuu, sts ? o1 : kC8NoStore);
                c8_st(qu, rqu, sts ? o1 : kC8NoStore);
            }
            // regularised blocks (util.cpp:135-143)
            double Qux_r[N], Quxo_r = Quxo, Quu_r = Quu;
#pragma unroll
            for (int i = 0; i < N; ++i) Qux_r[i] = Qux[i];
            if (a.reg == 2) {
                double bb = 0.0, bao = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    bb = fma(Bv[k], Bv[k], bb);
                    bao = fma(s.a[k], Bv[k], bao);
                }
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double ba = 0.0, ac[N];
                    get_col(ac, i);
#pragma unroll
                    for (int k = 0; k < N; ++k) ba = fma(ac[k], Bv[k], ba);
                    Qux_r[i] = fma(lam, ba, Qux[i]);
                }
                Quxo_r = fma(lam, bao, Quxo);
                Quu_r = fma(lam, bb, Quu);
            } else if (a.reg == 1) {
                Quu_r = Quu + lam;
            }
            // this step's A, B (Cxu) are dead from here on: the next step's (landed a step ago) go out now, to be read
            // at the end of the step (B) and in the next one (A)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            put_inputs(nx);
            // PD test (util.cpp:145-148): the same value in the eight lanes of a problem
            const bool bad = !(Quu_r - 100.0 * kEps * fabs(Quu_r) > 0.0);
            sdiv = (alive && bad) ? t : sdiv;
            alive = alive && !bad;
            const double nr = -rcp_nr(Quu_r);
            double K[N];
#pragma unroll
            for (int i = 0; i < N; ++i) K[i] = nr * Qux_r[i];   // util.cpp:150-152
            const double Ko = nr * Quxo_r, kff = nr * qu;
            const double Qk = Quu * kff;                        // unregularised Quu from here on (util.cpp:154-160)
            acc0 = alive ? fma(kff, qu, acc0) : acc0;
            acc1 = alive ? fma(0.5 * kff, Qk, acc1) : acc1;
            const double vno = fma(Quxo, kff, fma(Ko, Qk + qu, qxo));   // util.cpp:156-157, own element
            const double QKo = fma(Quu, Ko, Quxo);
            double Wo[N];                                                 // util.cpp:159-160, own column (halved)
#pragma unroll
            for (int i = 0; i < N; ++i) Wo[i] = 0.5 * fma(Qux[i], Ko, fma(K[i], QKo, Qxxo[i]));   // halved: V = W/2 + W'/2
            c8_st(Ko, rK, (alive && col_lane) ? oK : kC8NoStore);
            c8_st(kff, rkff, (alive && s_lane) ? okff : kC8NoStore);
            oK -= sK;
            okff -= skff;
            if constexpr (GN) {
                const double g = fabs(kff) / (fabs(s.ur) + 1.0);
                gm = alive ? nanmax(gm, g) : gm;
            }
            // exchange: W and v to every lane of the problem
#pragma unroll
            for (int i = 0; i < N / 2; ++i) wq2[(N * jw) / 2 + i] = c8_d2{Wo[2 * i], Wo[2 * i + 1]};
            wq[oV_ + jw] = vno;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double Wf[NN];
#pragma unroll
            for (int e = 0; e < NN / 2; ++e) {
                const c8_d2 x = xq2[e];
                Wf[2 * e] = x[0];
                Wf[2 * e + 1] = x[1];
            }
            get_inputs(Bv, Xv);
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                const c8_d2 x = xq2[oV_ / 2 + i];
                v[2 * i] = x[0];
                v[2 * i + 1] = x[1];
            }
            if constexpr (FULL) {
                const bool st = alive && col_lane;
                const unsigned oV = (unsigned)(((size_t)t * NN + N * jc) * NP * 8) + p8;
#pragma unroll
                for (int i = 0; i < N; ++i)   // column jc of (W + W') / 2: W(jc, i) from the exchange buffer
                    c8_st(Wo[i] + xq[N * i + jc], rV, st ? oV + (unsigned)i * np8 : kC8NoStore);
                c8_st(vno, rv, st ? (unsigned)(((size_t)t * N + jc) * NP * 8) + p8 : kC8NoStore);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int i = 0; i <= k; ++i) {   // util.cpp:161 (the halves are exact, so this is (W + W') / 2 bit for bit)
                    const double x = Wf[i + N * k] + Wf[k + N * i];
                    V[i + N * k] = x;
                    V[k + N * i] = x;
                }
        };

        // inputs run two steps ahead of the arithmetic (a step is ~3000 cycles, an HBM round trip under load fewer);
        // three ahead cost more in register traffic (the sets spill to AGPRs) than the extra slack was worth
        int t = T - 1;
        auto ahead = [&](Col8In<N>& s) { fetch(s, t >= 2 ? t - 2 : 0); };
        while (t >= 0) {
            ahead(s2);
            step(s0, s1, t);
            if (--t < 0) break;
            ahead(s0);
            step(s1, s2, t);
            if (--t < 0) break;
            ahead(s1);
            step(s2, s0, t);
            --t;
        }

        bool again = false;
        if (!frozen) {
            div = sdiv;
            dV0 = acc0;
            dV1 = acc1;
            gmax = gm;
            if (a.retry && sdiv >= 0) {  // ilqr.py:166-169
                dlam = fmax(dlam * a.sched.mult_lmbda, a.sched.mult_lmbda);
                lam = fmax(lam * dlam, a.sched.min_lmbda);
                again = (lam <= a.sched.max_lmbda) && (lam < 1e300);
            }
        }
        frozen = frozen || !again;
        // the wave repeats the sweep while any of its problems may still grow lambda; an exit every wave reaches
        if (__builtin_amdgcn_ballot_w64(again) == 0 || ++retries >= kMaxLambdaRetries) break;
    }
    if (j == 0 && !skip) {
        io.diverge[p] = div;
        io.dV[p] = dV0;
        io.dV[NP + p] = dV1;
        if (a.retry) {
            io.lmbda[p] = lam;
            if (io.dlmbda) io.dlmbda[p] = dlam;
        }
        if constexpr (GN) {
            if (io.g_norm) io.g_norm[p] = gmax;
        }
    }
}

// Shapes this family takes: dx = 6, du = 1, PROBLEM_MINOR (checked by the dispatcher), every array below 4 GB (32-bit
// byte offsets, and an out-of-range offset must exist for the dropped stores).
bool col8_shape_ok(int n, int m, int T, int64_t NP) {
    if (n != 6 || m != 1 || T < 1 || NP < 1) return false;
    const double lim = 4294967296.0 - 1024.0, slices = (double)(T + 1) * (double)NP * 8.0;
    return (double)n * n * slices < lim && (double)TRAJOPT_GAIN_BANKS * n * slices < lim;
}

template <bool SC>
static void col8_launch(const Col8Args& a, bool full, bool gn, hipStream_t st) {
    const dim3 grid((unsigned)((a.NP + 7) / 8)), block(64);
    if (full) {
        if (gn) hipLaunchKernelGGL((bwd_col8_kernel<6, true, true, SC>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bwd_col8_kernel<6, true, false, SC>), grid, block, 0, st, a);
    } else {
        if (gn) hipLaunchKernelGGL((bwd_col8_kernel<6, false, true, SC>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bwd_col8_kernel<6, false, false, SC>), grid, block, 0, st, a);
    }
}

int launch_backward_col8(const trajopt_backward_io_t& io, int reg, int n, int m, int T, int64_t NP, int retry,
                         const trajopt_lambda_schedule_t& sched, hipStream_t st, bool* handled) {
    *handled = false;
    if (!col8_shape_ok(n, m, T, NP)) return TRAJOPT_OK;
    *handled = true;
    Col8Args a;
    a.io = io; a.sched = sched; a.NP = (size_t)NP; a.T = T; a.reg = reg; a.retry = retry;
    const bool full = io.Qxx || io.Qux || io.Quu || io.qx || io.qu || io.V || io.v;
    const bool gn = io.uref != nullptr && io.g_norm != nullptr;
    if (io.cost_w) col8_launch<true>(a, full, gn, st);
    else col8_launch<false>(a, full, gn, st);
    TRAJOPT_HIP_CHECK(hipGetLastError());
    return TRAJOPT_OK;
}

}  // namespace trajopt
