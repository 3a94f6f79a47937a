// Developer microbenchmark: dependent-chain latency and issue interval of fp64 FMA / div / sincos, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s\n", (int)e_, #x); return; } } while (0)
template <int ILP, int OP>
__global__ void chain(double* out, int iters, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i + 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (OP == 0) x[i] = fma(x[i], a, b);
                else if (OP == 1) x[i] = b / x[i] + a;
                else if (OP == 2) { double s, c; sincos(x[i], &s, &c); x[i] = s + c * a; }
                else if (OP == 3) x[i] = sqrt(x[i]) + a;
                else if (OP == 4) x[i] = x[i] * a + b * x[i];   // mul + fma
                else if (OP == 5) x[i] = __builtin_amdgcn_rcp(x[i]) + a;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP, int OP>
void run(int waves_per_cu, const char* name, int iters) {
    double* out; CK(hipMalloc(&out, 1 << 24));
    dim3 grid(256), block(64 * waves_per_cu);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    chain<ILP, OP><<<grid, block>>>(out, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e0));
    chain<ILP, OP><<<grid, block>>>(out, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)iters * 16 * ILP;
    double ns = ms * 1e6 / ops;
    printf("%-8s ILP=%2d waves/CU=%2d: %.2f ns per op per wave (%.1f cyc @2.4GHz)\n", name, ILP, waves_per_cu, ns, ns * 2.4);
    CK(hipFree(out));
}
int main() {
    run<1, 0>(4, "fma", 4000); run<2, 0>(4, "fma", 4000); run<4, 0>(4, "fma", 4000); run<8, 0>(4, "fma", 4000);
    run<1, 0>(8, "fma", 4000); run<1, 0>(16, "fma", 4000); run<4, 0>(16, "fma", 4000);
    run<1, 4>(4, "mul+fma", 4000);
    run<1, 1>(4, "div", 1000); run<4, 1>(4, "div", 1000);
    run<1, 5>(4, "rcp", 1000);
    run<1, 3>(4, "sqrt", 1000);
    run<1, 2>(4, "sincos", 500); run<4, 2>(4, "sincos", 500);
    return 0;
}
