// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (ns per wave-instruction per SIMD at 8 waves per SIMD), with the
// operand patterns the compositing passes would use: accumulate pairs (C0, C1) += (w, w) * (c0, c1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b) {
    v2f acc[8];
    float s[16];
    for (int i = 0; i < 8; ++i) acc[i] = v2f{threadIdx.x * 1e-3f + i, 1.0f + i};
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 1e-3f + i;
    v2f c = {a, b}, w = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(w), "v"(c));  // broadcast w.x to both halves
                if (KIND == 2) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[2 * i]) : "v"(w.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[2 * i + 1]) : "v"(w.x), "v"(c.y)); }
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(c));
                if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(c));
            }
    }
    float t = 0;
    for (int i = 0; i < 8; ++i) t += acc[i].x + acc[i].y;
    for (int i = 0; i < 16; ++i) t += s[i];
    if (t == 1234.5f) out[0] = t;
}
template <int KIND> void run(const char *nm, float *d, int per) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 256 * 4;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, d, iters, 1.0001f, 0.9999f);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    const double inst = 8.0 * iters * 32 * per;  // wave-instructions per SIMD
    printf("%-44s %.3f ms  %.3f ns per instruction per SIMD  (%.2f ns per scalar FMA-equivalent)\n", nm, ms, ms * 1e6 / inst, ms * 1e6 / (8.0 * iters * 32 * 2));
}
int main() {
    float *d; hipMalloc(&d, 64);
    run<2>("2 x v_fma_f32", d, 2);
    run<0>("v_pk_fma_f32", d, 1);
    run<1>("v_pk_fma_f32 op_sel_hi broadcast", d, 1);
    run<3>("v_pk_mul_f32", d, 1);
    run<4>("v_pk_add_f32", d, 1);
    return 0;
}
