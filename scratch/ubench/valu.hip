// VALU issue-rate microbenchmark: wave64 v_fma_f32 throughput per SIMD vs resident waves and ILP.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP>
__global__ void k_fma(float *out, int iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}
template <int ILP>
__global__ void k_exp(float *out, int iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_amdgcn_exp2f(x[i] * a);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}
template <typename F>
double timeit(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float *d; hipMalloc(&d, 4);
    const int iters = 20000;
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock attr kHz %d\n", clk);
    for (int wpsimd : {1, 2, 4, 8}) {
        const int blocks = 256 * wpsimd;  // 256-thread blocks: 4 waves = one per SIMD
        auto run = [&](auto kern, int ilp, const char *nm) {
            double ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); });
            double inst_per_simd = (double)iters * 8 * ilp * wpsimd;  // wave-instructions per SIMD
            printf("%-8s ilp=%d waves/SIMD=%d  %.3f ms  -> %.2f cycles/wave-instr/SIMD @2.4GHz\n", nm, ilp, wpsimd, ms,
                   ms * 1e-3 * 2.4e9 / inst_per_simd);
        };
        run(k_fma<1>, 1, "fma"); run(k_fma<2>, 2, "fma"); run(k_fma<4>, 4, "fma"); run(k_fma<8>, 8, "fma");
        run(k_exp<4>, 4, "mul+exp");
    }
    return 0;
}
