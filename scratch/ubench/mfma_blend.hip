// Does the rank-1 colour update of the compositing forward, C[pixel][ch] += w[pixel] * c[ch] (four v_fma per 8x8 pass),
// run cheaper as ONE v_mfma_f32_4x4x1f32 (64 x 4 outer product on the matrix pipe, issued beside the VALU stream)?
// (1) layout check of the instruction, (2) time per pass of a loop shaped like k_blend_fwd_parts' pass, VALU vs MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float *w, const float *c, float *out) {
    // A: lane l supplies w[l]; B: lane l supplies c[l % 4]; D: four VGPRs per lane
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[threadIdx.x], c[threadIdx.x & 3], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = acc[r];
}

typedef float v2f __attribute__((ext_vector_type(2)));
// KIND 0: four v_fma; 1: one MFMA; 2: two v_pk_fma_f32 written as vector arithmetic; 3: two v_pk_fma_f32 as inline asm with op_sel broadcast of w
template <int KIND>
__global__ __launch_bounds__(512) void k_pass(float *out, int entries, const float4 *recs) {
    __shared__ float4 sh[3 * 64];
    __shared__ float shc[64 * 4];
    for (int i = threadIdx.x; i < 192; i += 512) sh[i] = recs[i];
    for (int i = threadIdx.x; i < 256; i += 512) shc[i] = 0.001f * (float)(i & 3) + 0.25f;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, lx = lane & 7u, ly = lane >> 3;
    float T[4] = {1, 1, 1, 1}, Cr[4] = {0, 0, 0, 0}, Cg[4] = {0, 0, 0, 0}, Cb[4] = {0, 0, 0, 0}, Dm[4] = {0, 0, 0, 0};
    v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v2f P0[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, P1[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};  // (Cr, Cg), (Cb, D)
    const float fx0 = (float)lx, fx1 = (float)lx + 8.0f, fy0 = (float)ly;
    for (int e = 0; e < entries; ++e) {
        const int j = e & 63;
        const float4 q0 = sh[j], q1 = sh[64 + j], q2 = sh[128 + j];
        const float cv = KIND == 1 ? shc[4 * j + (lane & 3u)] : 0.0f;  // the colour vector laid across each lane quad
        const float dxc[2] = {fx0 - q0.x, fx1 - q0.x};
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const float dy = row ? fy0 + 8.0f - q0.y : fy0 - q0.y;
            const float bdy = q0.w * dy, cyy = (q1.x * dy) * dy;
#pragma unroll
            for (int col = 0; col < 2; ++col) {
                const int s = 2 * row + col;
                const float dx = dxc[col];
                const float t = q0.z * dx + bdy;
                const float G = __builtin_amdgcn_exp2f(t * dx + cyy);
                const float a1 = __builtin_amdgcn_fmed3f(G * q1.y, 0.0f, 1.0f);
                const float w = a1 * T[s];
                if (KIND == 1) {
                    acc[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, cv, acc[s], 0, 0, 0);
                } else if (KIND == 2) {
                    const v2f wv = {w, w};
                    P0[s] += wv * v2f{q1.z, q1.w}; P1[s] += wv * v2f{q2.x, q2.y};
                } else if (KIND == 3) {
                    v2f wv; wv.x = w;  // .y never read (op_sel_hi picks the low half twice)
                    const v2f c0 = {q1.z, q1.w}, c1 = {q2.x, q2.y};
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(P0[s]) : "v"(wv), "v"(c0));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(P1[s]) : "v"(wv), "v"(c1));
                } else {
                    Cr[s] += w * q1.z; Cg[s] += w * q1.w; Cb[s] += w * q2.x; Dm[s] += w * q2.y;
                }
                T[s] = fmaf(w, -0.99f, T[s]);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += T[i] + Cr[i] + Cg[i] + Cb[i] + Dm[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + P0[i].x + P0[i].y + P1[i].x + P1[i].y;
    if (s == 1234.5f) out[0] = s;
}

int main() {
    float *w, *c, *out; hipMalloc(&w, 256); hipMalloc(&c, 16); hipMalloc(&out, 1024);
    float hw[64], hc[4] = {1, 10, 100, 1000}, ho[256];
    for (int i = 0; i < 64; ++i) hw[i] = (float)(i + 1);
    hipMemcpy(w, hw, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, w, c, out);
    hipMemcpy(ho, out, 1024, hipMemcpyDeviceToHost);
    // expectation: lane l, VGPR r holds w[4 (l / 4) + r] * c[l % 4]
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (ho[l * 4 + r] != hw[4 * (l / 4) + r] * hc[l & 3]) ++bad;
    printf("layout: lane l VGPR r = w[4 (l/4) + r] * c[l %% 4]: %s (%d mismatches); lane 5: %g %g %g %g\n", bad ? "NO" : "yes", bad, ho[20], ho[21], ho[22], ho[23]);
    float4 *recs; hipMalloc(&recs, 192 * 16);
    float4 hr[192];
    for (int i = 0; i < 64; ++i) { hr[i] = {3.5f + i * 0.1f, 4.0f, -0.02f, -0.001f}; hr[64 + i] = {-0.02f, 0.8f, 0.3f, 0.6f}; hr[128 + i] = {0.9f, 2.0f, 0, 0}; }
    hipMemcpy(recs, hr, sizeof(hr), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int entries = 4096;
    for (int rep = 0; rep < 2; ++rep)
        for (int variant = 0; variant < 4; ++variant) {
            const int blocks = 256 * 4;  // 4 blocks of 8 waves per CU = 8 waves per SIMD
            for (int warm = 0; warm < 2; ++warm) {
                hipEventRecord(a);
                if (variant == 1) hipLaunchKernelGGL(k_pass<1>, dim3(blocks), dim3(512), 0, 0, out, entries, recs);
                else if (variant == 2) hipLaunchKernelGGL(k_pass<2>, dim3(blocks), dim3(512), 0, 0, out, entries, recs);
                else if (variant == 3) hipLaunchKernelGGL(k_pass<3>, dim3(blocks), dim3(512), 0, 0, out, entries, recs);
                else hipLaunchKernelGGL(k_pass<0>, dim3(blocks), dim3(512), 0, 0, out, entries, recs);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            // wave-passes per SIMD: 8 waves x entries x 4 passes
            static const char *names[4] = {"VALU (4 v_fma) colour update", "MFMA 4x4x1 colour update", "2 v_pk_fma_f32 (vector arithmetic)", "2 v_pk_fma_f32 (asm, op_sel broadcast)"};
            printf("%s: %.3f ms, %.2f ns per pass per SIMD (8 waves per SIMD)\n", names[variant], ms,
                   ms * 1e6 / (8.0 * entries * 4));
        }
    return 0;
}
