// Second pass of the gfx950 VALU issue model: select / compare / integer / min-max ops (inline asm).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND, int ILP>
__global__ void k(float *out, int iters, float a, float b, unsigned long long m) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
    const unsigned long long mask = m;
    unsigned long long sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(mask));
                if (KIND == 1) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(x[i]) : "s"(mask));
                if (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(a));
                if (KIND == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
                if (KIND == 7) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sink) : "v"(x[i]), "v"(a));
                if (KIND == 8) asm volatile("v_cmp_lt_u32_sdwa vcc, %0, %1 src0_sel:DWORD src1_sel:WORD_1" : : "v"(x[i]), "v"(a) : "vcc");
                if (KIND == 9) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
                if (KIND == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "s"(a));
                if (KIND == 13) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]));
                if (KIND == 14) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 15) asm volatile("v_fma_f32 %0, -%0, %1, %0" : "+v"(x[i]) : "v"(a));
                if (KIND == 16) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
                if (KIND == 18) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f || sink == 77) out[0] = s;
}
template <typename F>
double timeit(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
template <int KIND, int ILP>
void run(const char *nm, float *d, unsigned long long m = 0x5555555555555555ull, float a = 1.0001f) {
    const int iters = 4000;
    printf("%-28s ilp=%d :", nm, ILP);
    for (int wpsimd : {1, 4, 8}) {
        const int blocks = 256 * wpsimd;
        double ms = timeit([&] { hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, a, 0.5f, m); });
        double inst_per_simd = (double)iters * 8 * ILP * wpsimd;
        printf("  w%d %.2f", wpsimd, ms * 1e-3 * 2.4e9 / inst_per_simd);
    }
    printf("\n");
}
int main() {
    float *d; hipMalloc(&d, 4);
    run<0, 4>("cndmask e64 sgpr mask 0x55", d);
    run<0, 4>("cndmask e64 sgpr mask all1", d, ~0ull);
    run<0, 4>("cndmask e64 sgpr mask 0", d, 0ull);
    run<1, 4>("cndmask e64 (0, x) 0x55", d);
    run<17, 4>("cndmask e32 vcc", d);
    run<11, 4>("cmp+cndmask (per pair)", d);
    run<2, 4>("v_and_b32", d); run<18, 4>("v_bfi_b32", d);
    run<3, 4>("v_max_f32", d); run<14, 4>("v_min_f32", d);
    run<4, 4>("v_add_u32", d); run<5, 4>("v_mov_b32", d);
    run<6, 4>("v_cmp_lt_f32 vcc", d); run<7, 4>("v_cmp_lt_f32 e64 sgpr", d); run<8, 4>("v_cmp_lt_u32 sdwa", d);
    run<9, 4>("v_fmac_f32", d); run<10, 4>("v_sub_f32", d); run<12, 4>("v_mul_f32 sgpr src", d);
    run<13, 4>("v_cvt_f32_u32", d); run<15, 4>("v_fma_f32 neg", d); run<16, 4>("v_mul_legacy", d);
    run<9, 1>("v_fmac_f32", d);
    run<3, 4>("v_max_f32 (a=denormal)", d, 0, 1e-40f);
    run<9, 4>("v_fmac_f32 (a=denormal)", d, 0, 1e-40f);
    return 0;
}
