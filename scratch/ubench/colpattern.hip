// What the column kernels' access pattern can deliver: 805 MB of float2 planes [planes][512][512] read once by blocks of 1024
// threads that each take a (512 rows x SEG bytes) column tile of one plane after the other (16 planes per block, like
// k_colfft_fwd), nothing else: loads are independent, summed into one register.  SEG = 64 / 128 / 256 bytes per row segment,
// and for comparison the same bytes read as whole contiguous rows.  Also the mirrored store pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int H = 512, W = 512, P = 16, IMG = 8, CH = 3;
template <int TC, bool WRITE>
__global__ __launch_bounds__(1024) void k_tile(float2 *field, float2 *out) {
    // grid (W / TC, CH, IMG); thread: col = tid % TC, r0 = tid / TC; rows r0 + k * (1024 / TC)
    const int col = threadIdx.x % TC, r0 = threadIdx.x / TC, RS = 1024 / TC;
    const int c = blockIdx.y, b = blockIdx.z, c0 = blockIdx.x * TC;
    float2 s = make_float2(0.f, 0.f);
    for (int p = 0; p < P; ++p) {
        float2 *f = field + (((size_t)b * P + p) * CH + c) * H * W + c0 + col;
#pragma unroll
        for (int r = r0; r < H; r += RS) {
            if (WRITE) f[(size_t)r * W] = make_float2((float)p, (float)r);
            else { const float2 v = f[(size_t)r * W]; s.x += v.x; s.y += v.y; }
        }
    }
    if (!WRITE && s.x == 123.456f) out[0] = s;
}
template <bool WRITE>
__global__ __launch_bounds__(1024) void k_rows(float2 *field, float2 *out) {  // same bytes per block, contiguous: 16 rows x 512 points per step
    const size_t per_block = (size_t)H * 16 * P;  // elements
    float2 *f = field + (size_t)blockIdx.x * per_block;
    float2 s = make_float2(0.f, 0.f);
    for (size_t i = threadIdx.x; i < per_block; i += 1024) {
        if (WRITE) f[i] = make_float2(1.f, 2.f);
        else { const float2 v = f[i]; s.x += v.x; s.y += v.y; }
    }
    if (!WRITE && s.x == 123.456f) out[0] = s;
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
    printf("%-34s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / 20, bytes / (ms / 20 * 1e-3) / 1e12); fflush(stdout);
}
int main() {
    float2 *d, *o; const size_t n = (size_t)IMG * P * CH * H * W;
    hipMalloc(&d, n * 8); hipMalloc(&o, 64); hipMemset(d, 0, n * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run("read  tiles 512 x  64 B", [&] { hipLaunchKernelGGL((k_tile<8, false>), dim3(W / 8, CH, IMG), dim3(1024), 0, 0, d, o); });
        run("read  tiles 512 x 128 B", [&] { hipLaunchKernelGGL((k_tile<16, false>), dim3(W / 16, CH, IMG), dim3(1024), 0, 0, d, o); });
        run("read  tiles 512 x 256 B", [&] { hipLaunchKernelGGL((k_tile<32, false>), dim3(W / 32, CH, IMG), dim3(1024), 0, 0, d, o); });
        run("read  tiles 512 x 512 B", [&] { hipLaunchKernelGGL((k_tile<64, false>), dim3(W / 64, CH, IMG), dim3(1024), 0, 0, d, o); });
        run("read  contiguous", [&] { hipLaunchKernelGGL((k_rows<false>), dim3(W / 16 * CH * IMG), dim3(1024), 0, 0, d, o); });
        run("write tiles 512 x  64 B", [&] { hipLaunchKernelGGL((k_tile<8, true>), dim3(W / 8, CH, IMG), dim3(1024), 0, 0, d, o); });
        run("write tiles 512 x 128 B", [&] { hipLaunchKernelGGL((k_tile<16, true>), dim3(W / 16, CH, IMG), dim3(1024), 0, 0, d, o); });
        run("write tiles 512 x 256 B", [&] { hipLaunchKernelGGL((k_tile<32, true>), dim3(W / 32, CH, IMG), dim3(1024), 0, 0, d, o); });
        run("write contiguous", [&] { hipLaunchKernelGGL((k_rows<true>), dim3(W / 16 * CH * IMG), dim3(1024), 0, 0, d, o); });
    }
    return 0;
}
