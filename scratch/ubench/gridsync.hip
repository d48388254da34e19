// Cost of a cooperative-groups grid barrier on gfx950 (512 blocks x 256 threads), vs the gap between two launches.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
namespace cg = cooperative_groups;
__global__ void k_sync(int n, unsigned *out) {
    cg::grid_group g = cg::this_grid();
    unsigned acc = 0;
    for (int i = 0; i < n; ++i) { acc += blockIdx.x + i; g.sync(); }
    if (threadIdx.x == 0 && acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ void k_empty(unsigned *out) { if (threadIdx.x == 999) out[0] = 1; }
int main() {
    unsigned *d; hipMalloc(&d, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {64, 512, 1024}) {
        for (int n : {0, 12, 48}) {
            void *args[] = {&n, &d};
            hipError_t e = hipLaunchCooperativeKernel((void *)k_sync, dim3(blocks), dim3(256), args, 0, 0);
            if (e != hipSuccess) { printf("coop launch failed: %s\n", hipGetErrorString(e)); return 1; }
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 20; ++r) hipLaunchCooperativeKernel((void *)k_sync, dim3(blocks), dim3(256), args, 0, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("blocks %4d  syncs %2d : %.2f us per launch\n", blocks, n, ms * 1e3 / 20); fflush(stdout);
        }
    }
    hipEventRecord(a);
    for (int r = 0; r < 240; ++r) hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, 0, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("empty kernel back-to-back: %.2f us per launch\n", ms * 1e3 / 240);
    return 0;
}
