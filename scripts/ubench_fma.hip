// Developer microbenchmark: fp64 FMA issue rate per SIMD as a function of waves/SIMD and ILP.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void fma_chain(double* out, int iters, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
void run(int waves_per_cu, const char* name) {
    double* out; hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    dim3 grid(256), block(64 * waves_per_cu);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fma_chain<ILP><<<grid, block>>>(out, iters, 1.0000001, 1e-9);
    hipEventRecord(e0);
    fma_chain<ILP><<<grid, block>>>(out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fmas_per_wave = (double)iters * ILP;
    double ns_per_fma = ms * 1e6 / fmas_per_wave;   // per wave-instruction on the critical wave
    printf("%s ILP=%d waves/CU=%d: %.3f ms, %.2f ns per wave-FMA (%.1f cycles @2.4GHz), chip %.1f TFLOP/s\n", name, ILP,
           waves_per_cu, ms, ns_per_fma, ns_per_fma * 2.4, 2.0 * 64 * fmas_per_wave * waves_per_cu * 256 / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(4, "f64"); run<2>(4, "f64"); run<4>(4, "f64"); run<8>(4, "f64"); run<16>(4, "f64");
    run<8>(8, "f64"); run<8>(16, "f64"); run<16>(1, "f64"); run<1>(1, "f64");
    return 0;
}
