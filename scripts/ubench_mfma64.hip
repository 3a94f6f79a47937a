// Developer microbenchmark: v_mfma_f64_16x16x4_f64 operand layout check and issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s\n", (int)e_, #x); return 1; } } while (0)

__global__ void layout_check(const double* A, const double* B, double* D) {  // A 16x4, B 4x16 row-major; D 16x16 row-major
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int CHAINS>
__global__ void rate(double* out, int iters) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 c[CHAINS];
    for (int k = 0; k < CHAINS; ++k) c[k] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < CHAINS; ++k) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
    }
    double s = 0;
    for (int k = 0; k < CHAINS; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
    out[blockIdx.x * blockDim.x + l] = s;
}

template <int CHAINS>
int run(int waves_per_cu, int iters) {
    double* out; CK(hipMalloc(&out, 1 << 24));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate<CHAINS><<<256, 64 * waves_per_cu>>>(out, iters);
    CK(hipEventRecord(e0));
    rate<CHAINS><<<256, 64 * waves_per_cu>>>(out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * 8 * CHAINS;
    printf("mfma_f64_16x16x4 chains=%d waves/CU=%d: %.1f ns per MFMA per wave (%.1f cyc @2.4GHz), chip %.1f TFLOP/s\n", CHAINS,
           waves_per_cu, ms * 1e6 / n, ms * 1e6 / n * 2.4, 2048.0 * n * waves_per_cu * 256 / ms / 1e9);
    CK(hipFree(out));
    return 0;
}

int main() {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.37; B[i] = 2 - i * 0.11 + (i % 5); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout_check<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("layout check: max |D - A*B| = %.3e (%s)\n", err, err < 1e-9 ? "layout as documented" : "LAYOUT MISMATCH");
    if (run<1>(4, 2000)) return 1; if (run<2>(4, 2000)) return 1; if (run<4>(4, 2000)) return 1;
    if (run<1>(8, 2000)) return 1; if (run<4>(8, 2000)) return 1; if (run<2>(16, 2000)) return 1;
    return 0;
}
