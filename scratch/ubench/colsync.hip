// The column kernels' memory pipeline without their arithmetic: blocks of 1024 threads (one per CU: 132 KB of LDS), 16 planes per
// block, per plane one 64 KB tile (8 x 8 bytes per thread), a block barrier per plane, DEPTH tiles requested ahead (registers).
// TILE: 512 rows x 128 bytes at a 4 KB stride (the real pattern); CONT: the same 64 KB as one contiguous chunk.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int H = 512, W = 512, P = 16, IMG = 8, CH = 3;
template <bool TILE, int DEPTH, int MODE>  // MODE bit 0: LDS round trip, bit 1: barriers
__global__ __launch_bounds__(1024) void k(const float2 *field, float2 *out) {
    __shared__ float2 lds[132 * 1024 / 8];
    const int col = threadIdx.x % 16, r0 = threadIdx.x / 16;
    const int c = blockIdx.y, b = blockIdx.z, c0 = blockIdx.x * 16;
    auto addr = [&](int p, int e) -> const float2 * {
        const float2 *pl = field + (((size_t)b * P + p) * CH + c) * H * W;
        return TILE ? pl + (size_t)(r0 + 64 * e) * W + c0 + col : pl + (size_t)blockIdx.x * 8192 + e * 1024 + threadIdx.x;
    };
    float2 buf[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int e = 0; e < 8; ++e) buf[d][e] = *addr(d, e);
    float2 s = make_float2(0.f, 0.f);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float2 cur[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cur[e] = buf[p % DEPTH][e];
        if (p + DEPTH < P) {
#pragma unroll
            for (int e = 0; e < 8; ++e) buf[p % DEPTH][e] = *addr(p + DEPTH, e);
        }
        if (MODE & 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) lds[(r0 + 64 * e) * 16 + col] = cur[e];
        }
        if (MODE & 2) __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float2 v = (MODE & 1) ? lds[(8 * r0 + e) * 16 + col] : cur[e]; s.x += v.x; s.y += v.y; }
        if (MODE & 2) __syncthreads();
    }
    if (s.x == 123.456f) out[0] = lds[threadIdx.x + 8000];
}
__global__ __launch_bounds__(1024) void k_fill(float2 *field, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) { unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; field[i] = make_float2(__uint_as_float(0x3f800000u | (h >> 9)), __uint_as_float(0x3f800000u | ((h * 3266489917u) >> 9))); }
}
template <class F>
void run(const char *name, F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)IMG * P * CH * H * W * 8;
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / 20, bytes / (ms / 20 * 1e-3) / 1e12); fflush(stdout);
}
int main() {
    float2 *d, *o; const size_t n = (size_t)IMG * P * CH * H * W;
    hipMalloc(&d, n * 8); hipMalloc(&o, 64); hipMemset(d, 0, n * 8);
    const dim3 g(W / 16, CH, IMG);
    run("tile, 1 ahead, LDS + barriers", [&] { hipLaunchKernelGGL((k<true, 1, 3>), g, dim3(1024), 0, 0, d, o); });
    run("tile, 2 ahead, LDS + barriers", [&] { hipLaunchKernelGGL((k<true, 2, 3>), g, dim3(1024), 0, 0, d, o); });
    run("tile, 4 ahead, LDS + barriers", [&] { hipLaunchKernelGGL((k<true, 4, 3>), g, dim3(1024), 0, 0, d, o); });
    run("contiguous, 1 ahead, LDS + barriers", [&] { hipLaunchKernelGGL((k<false, 1, 3>), g, dim3(1024), 0, 0, d, o); });
    run("tile, 1 ahead, barriers only", [&] { hipLaunchKernelGGL((k<true, 1, 2>), g, dim3(1024), 0, 0, d, o); });
    run("tile, 1 ahead, LDS only (racy)", [&] { hipLaunchKernelGGL((k<true, 1, 1>), g, dim3(1024), 0, 0, d, o); });
    run("tile, 1 ahead, neither", [&] { hipLaunchKernelGGL((k<true, 1, 0>), g, dim3(1024), 0, 0, d, o); });
    run("tile, 4 ahead, neither", [&] { hipLaunchKernelGGL((k<true, 4, 0>), g, dim3(1024), 0, 0, d, o); });
    run("tile, 4 ahead, barriers only", [&] { hipLaunchKernelGGL((k<true, 4, 2>), g, dim3(1024), 0, 0, d, o); });
    run("contiguous, 4 ahead, neither", [&] { hipLaunchKernelGGL((k<false, 4, 0>), g, dim3(1024), 0, 0, d, o); });
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(1024), 0, 0, d, n); hipDeviceSynchronize();
    run("random data: tile 1 ahead LDS + barriers", [&] { hipLaunchKernelGGL((k<true, 1, 3>), g, dim3(1024), 0, 0, d, o); });
    run("random data: tile 2 ahead LDS + barriers", [&] { hipLaunchKernelGGL((k<true, 2, 3>), g, dim3(1024), 0, 0, d, o); });
    run("fill only (row-FFT stand-in)", [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(1024), 0, 0, d, n); });
    run("fill + tile 1 ahead LDS + barriers", [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(1024), 0, 0, d, n); hipLaunchKernelGGL((k<true, 1, 3>), g, dim3(1024), 0, 0, d, o); });
    run("fill + tile 2 ahead LDS + barriers", [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(1024), 0, 0, d, n); hipLaunchKernelGGL((k<true, 2, 3>), g, dim3(1024), 0, 0, d, o); });
    return 0;
}
