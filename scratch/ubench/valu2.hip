// VALU issue model on gfx950, with inline asm so the compiler cannot repack:
// cycles per wave-instruction per SIMD for plain / packed / transcendental / DPP ops vs ILP and resident waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND, int ILP>
__global__ void k(float *out, int iters, float a, float b) {
    float x[8]; v2f y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = v2f{x[i], x[i] + 1}; }
    v2f a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                if (KIND == 4) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
                if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
                if (KIND == 7) {  // one exp followed by 3 independent plain fma: does the transcendental overlap?
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i + 1) & 7]) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i + 2) & 7]) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i + 3) & 7]) : "v"(a), "v"(b));
                }
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
                if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 10) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x[i]));
                if (KIND == 11) asm volatile("s_and_b64 vcc, vcc, exec\n v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <typename F>
double timeit(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
template <int KIND, int ILP>
void run(const char *nm, float *d, int mult = 1) {
    const int iters = 4000;
    printf("%-22s ilp=%d :", nm, ILP);
    for (int wpsimd : {1, 2, 4, 8}) {
        const int blocks = 256 * wpsimd;
        double ms = timeit([&] { hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); });
        double inst_per_simd = (double)iters * 8 * ILP * wpsimd * mult;
        printf("  w%d %.2f", wpsimd, ms * 1e-3 * 2.4e9 / inst_per_simd);
    }
    printf("   (cycles per wave-instr per SIMD @2.4GHz)\n");
}
int main() {
    float *d; hipMalloc(&d, 4);
    run<0, 1>("v_fma_f32", d); run<0, 2>("v_fma_f32", d); run<0, 4>("v_fma_f32", d); run<0, 8>("v_fma_f32", d);
    run<1, 1>("v_pk_fma_f32", d); run<1, 2>("v_pk_fma_f32", d); run<1, 4>("v_pk_fma_f32", d); run<1, 8>("v_pk_fma_f32", d);
    run<8, 1>("v_pk_mul_f32", d); run<8, 4>("v_pk_mul_f32", d);
    run<9, 1>("v_mul_f32", d); run<9, 4>("v_mul_f32", d);
    run<2, 1>("v_exp_f32", d); run<2, 4>("v_exp_f32", d);
    run<3, 1>("v_rcp_f32", d); run<3, 4>("v_rcp_f32", d);
    run<4, 1>("v_med3_f32", d); run<4, 4>("v_med3_f32", d);
    run<6, 1>("v_cndmask_b32", d); run<6, 4>("v_cndmask_b32", d);
    run<5, 1>("v_add_dpp row_shr", d); run<5, 4>("v_add_dpp row_shr", d);
    run<10, 1>("v_add_dpp row_bcast", d); run<10, 4>("v_add_dpp row_bcast", d);
    run<7, 4>("exp+3fma (per instr)", d, 4);
    run<11, 4>("s_and+fma (per pair)", d);
    return 0;
}
