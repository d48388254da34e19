// buffer_load ... lds on gfx950 (16 bytes per lane): where does lane i's data land?  One block of 256 threads (4 waves); wave w loads
// 64 x 16 B starting at global float4 index 64 w + lane (+ soffset) into LDS at base_w; afterwards every thread copies LDS -> out.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float4 *in, float4 *out) {
    __shared__ float4 buf[256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, 0x7fffffff, 0x00020000);
    // lanes load in REVERSED order within the wave (global index 64 w + 63 - lane): the LDS slot tells whether placement follows the lane id
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(uintptr_t)(buf + 64 * wave), 16, (63 - lane) * 16, wave * 64 * 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) (and everything else)
    __syncthreads();
    out[threadIdx.x] = buf[threadIdx.x];
}
int main() {
    float4 h[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = make_float4((float)i, 0, 0, 0);
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o);
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; i += 21) printf("lds slot %3d holds global element %3.0f\n", i, h[i].x);
    int ok = 1;
    for (int i = 0; i < 256; ++i) ok &= (h[i].x == (float)((i / 64) * 64 + 63 - (i % 64)));
    printf("placement = base + lane * 16 : %s\n", ok ? "yes" : "NO");
    return 0;
}
