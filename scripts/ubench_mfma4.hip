// Developer microbenchmark: v_mfma_f64_4x4x4_4b_f64 operand layout discovery, dependent-chain latency, and the
// 32-bit DPP controls (quad_perm / row_shr / row_newbcast) the 4-problems-per-wave backward kernel relies on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s\n", (int)e_, #x); return 1; } } while (0)

// mask[la * 64 + lb] = lanes whose D is non-zero when only lane la feeds A = 1 and only lane lb feeds B = 1
__global__ void discover(unsigned long long* mask) {
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (l == 0) mask[la * 64 + lb] = m;
        }
}

template <int MODE>  // 0: D -> C chain, 1: D -> B chain, 2: D -> A chain
__global__ void chain(double* out, int iters) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-9, b = 1e-3, c = 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
            if (MODE == 1) b = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
            if (MODE == 2) a = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + l] = a + b + c;
}

template <int CTRL>
__device__ __forceinline__ double dpp64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void dpp_check(double* out) {
    const int l = threadIdx.x;
    const double x = (double)l;
    out[0 * 64 + l] = dpp64<0x150 + 5>(x);                 // row_newbcast:5  -> 16 * (l / 16) + 5
    out[1 * 64 + l] = dpp64<0x00>(x);                      // quad_perm:[0,0,0,0] -> 4 * (l / 4)
    out[2 * 64 + l] = dpp64<0x110 + 4>(x);                 // row_shr:4 -> l - 4 within the row (0 where none: old = 0)
    out[3 * 64 + l] = dpp64<0x120 + 4>(x);                 // row_ror:4 -> (l - 4) mod 16 within the row
    out[4 * 64 + l] = dpp64<0x55>(x);                      // quad_perm:[1,1,1,1]
}

int main() {
    unsigned long long* dm; CK(hipMalloc(&dm, 4096 * 8));
    discover<<<1, 64>>>(dm);
    std::vector<unsigned long long> m(4096);
    CK(hipMemcpy(m.data(), dm, 4096 * 8, hipMemcpyDeviceToHost));
    // hypothesis: block = lane / 16; A: i = lane % 4, k = (lane / 4) % 4; B: j = lane % 4, k = (lane / 4) % 4;
    //             D: j = lane % 4, i = (lane / 4) % 4
    int bad = 0, nz = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            unsigned long long want = 0;
            if (la / 16 == lb / 16 && (la / 4) % 4 == (lb / 4) % 4) want = 1ull << (16 * (la / 16) + 4 * (la % 4) + (lb % 4));
            if (m[la * 64 + lb]) ++nz;
            if (m[la * 64 + lb] != want) {
                if (bad < 12) printf("  la=%d lb=%d mask=%016llx want=%016llx\n", la, lb, m[la * 64 + lb], want);
                ++bad;
            }
        }
    printf("mfma_f64_4x4x4 layout: %d non-zero (la, lb) pairs, %d deviations from the hypothesis (%s)\n", nz, bad,
           bad ? "LAYOUT DIFFERS" : "A[i][k] @ 16b+4k+i, B[k][j] @ 16b+4k+j, D[i][j] @ 16b+4i+j");
    if (bad) {  // dump the actual structure for lanes of block 0
        for (int la = 0; la < 16; ++la)
            for (int lb = 0; lb < 64; ++lb)
                if (m[la * 64 + lb]) printf("  A-lane %d x B-lane %d -> D mask %016llx\n", la, lb, m[la * 64 + lb]);
    }
    double* out; CK(hipMalloc(&out, 1 << 22));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int mode = 0; mode < 3; ++mode)
        for (int w = 1; w <= 4; w *= 2) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) chain<0><<<256, 256 * w>>>(out, iters);
                if (mode == 1) chain<1><<<256, 256 * w>>>(out, iters);
                if (mode == 2) chain<2><<<256, 256 * w>>>(out, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("chain D->%c, %d wave(s)/SIMD: %.2f ns per MFMA per wave = %.1f cycles @2.4GHz\n", "CBA"[mode], w,
                   ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) * 2.4);
        }
    double* dd; CK(hipMalloc(&dd, 5 * 64 * 8));
    dpp_check<<<1, 64>>>(dd);
    std::vector<double> r(5 * 64);
    CK(hipMemcpy(r.data(), dd, 5 * 64 * 8, hipMemcpyDeviceToHost));
    const char* nm[5] = {"row_newbcast:5", "quad_perm:[0,0,0,0]", "row_shr:4", "row_ror:4", "quad_perm:[1,1,1,1]"};
    for (int k = 0; k < 5; ++k) {
        printf("%-20s:", nm[k]);
        for (int l = 0; l < 32; ++l) printf(" %2d", (int)r[k * 64 + l]);
        printf(" ...\n");
    }
    return 0;
}
