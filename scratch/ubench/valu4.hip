// Third pass of the gfx950 VALU issue model: SALU-written masks, constants, SGPR operands, LDS broadcast.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND, int ILP>
__global__ void k(float *out, int iters, float a, float b, unsigned long long m) {
    __shared__ float4 sh[64];
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
    sh[threadIdx.x & 63] = float4{a, b, a, b};
    __syncthreads();
    const unsigned long long mask = m;
    unsigned long long tmp;
    float4 q = {0, 0, 0, 0};
    uint32_t sidx = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) asm volatile("s_and_b64 vcc, %1, exec\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[i]) : "s"(mask) : "vcc", "scc");
                if (KIND == 1) asm volatile("s_and_b64 %1, %2, exec\n v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(x[i]), "=s"(tmp) : "s"(mask) : "scc");
                if (KIND == 2) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x[i]));
                if (KIND == 3) asm volatile("v_add_f32 %0, 0x41000000, %0" : "+v"(x[i]));
                if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
                if (KIND == 5) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 6) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 7) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]));
                if (KIND == 8) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %3\n v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %1, 0, %1, vcc\n v_cndmask_b32 %2, 0, %2, vcc"
                                            : "+v"(x[i]), "+v"(x[(i + 1) & 7]), "+v"(x[(i + 2) & 7]) : "v"(a) : "vcc");
                if (KIND == 10) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(a));
                if (KIND == 11) { asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(sidx)); asm volatile("s_waitcnt lgkmcnt(0)\n v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(q.x), "v"(q.y)); }
                if (KIND == 12) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sidx) : "v"(x[i]));
                if (KIND == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(a));
                if (KIND == 14) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 15) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, %2\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[i]) : "v"(a), "s"(mask) : "vcc", "scc");
                if (KIND == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 4\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
                if (KIND == 17) asm volatile("v_mul_f32 %0, 0x3f7d70a4, %0" : "+v"(x[i]));
                if (KIND == 18) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 19) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 20) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
            }
    }
    float s = q.x + q.z;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f || tmp == 77 || sidx == 1234567) out[0] = s;
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
    printf("%-36s ilp=%d :", nm, ILP); fflush(stdout);
    for (int wpsimd : {1, 4, 8}) {
        const int blocks = 256 * wpsimd;
        double ms = timeit([&] { hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, 0x5555555555555555ull); });
        double inst_per_simd = (double)iters * 8 * ILP * wpsimd * mult;
        printf("  w%d %.2f", wpsimd, ms * 1e-3 * 2.4e9 / inst_per_simd);
    }
    printf("\n"); fflush(stdout);
}
int main() {
    float *d; hipMalloc(&d, 4);
    run<0, 4>("s_and vcc + cndmask vcc (per pair)", d);
    run<1, 4>("s_and sgpr + cndmask e64 (per pair)", d);
    run<15, 4>("cmp + s_and vcc + cnd (per triple)", d);
    run<16, 4>("cmp + s_nop 4 + cnd (per triple)", d);
    run<9, 4>("cmp + 3 cnd vcc (per group of 4)", d);
    run<2, 4>("v_add_f32 inline const 1.0", d); run<3, 4>("v_add_f32 literal", d); run<17, 4>("v_mul_f32 literal", d);
    run<14, 4>("v_fma_f32 inline const", d);
    run<4, 4>("v_fma_f32 sgpr src", d); run<5, 4>("v_mul_f32_e64 neg", d); run<6, 4>("v_mul_f32_e64 abs", d);
    run<10, 4>("v_mul_f32_e64 clamp", d);
    run<7, 4>("v_lshlrev_b32", d); run<8, 4>("v_or_b32", d); run<18, 4>("v_xor_b32", d); run<19, 4>("v_sub_u32", d);
    run<13, 4>("v_mad_u32_u24", d); run<20, 4>("v_cmp_lt_u32 vcc", d);
    run<11, 4>("ds_read_b128 bcast + fmac (pair)", d);
    run<12, 4>("v_readfirstlane", d);
    return 0;
}
