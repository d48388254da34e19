// Importance-subsampling hand-off between the decoder and the rasterizer (reference
// scripts/training/train_gaussian_decoder.py:1160-1187): the K Gaussians drawn by torch.multinomial are gathered
// out of every per-Gaussian tensor of the batch.  The reference does this with six advanced-indexing ops (and
// autograd's six index_put backward ops); here ONE launch gathers all tensors of a Gaussian (14-17 floats), and one
// launch scatters the gradients back.  Pure data movement, HBM/L2-bound: 4*(14 + phase) bytes in + out per Gaussian.
#include "fgs_internal.h"

namespace {

struct GatherPtrs {
    const float *pos, *scale, *quat, *color, *opacity, *phase;
    float *o_pos, *o_scale, *o_quat, *o_color, *o_opacity, *o_phase;
};

// thread = one (image, selected Gaussian); SCATTER: the same walk, copying out -> in (gradients; indices are unique:
// sampling is without replacement, TGD:1173)
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_gather(int32_t total, int32_t n_in, int32_t n_out, int32_t phase_channels,
                                                const long long *__restrict__ indices, GatherPtrs p) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int32_t b = i / n_out, k = i - b * n_out;
    const long long src = indices[k];
    if (src < 0 || src >= n_in) return;  // out-of-range index: skipped (the host wrapper validates)
    const size_t in = (size_t)b * n_in + (size_t)src, out = (size_t)i;
    auto copy = [&](const float *a, float *o, int w) {
        if (!a || !o) return;
#pragma unroll 4
        for (int c = 0; c < w; ++c) {
            if (SCATTER) const_cast<float *>(a)[in * w + c] = o[out * w + c];
            else o[out * w + c] = a[in * w + c];
        }
    };
    copy(p.pos, p.o_pos, 3); copy(p.scale, p.o_scale, 3); copy(p.quat, p.o_quat, 4); copy(p.color, p.o_color, 3);
    copy(p.opacity, p.o_opacity, 1); copy(p.phase, p.o_phase, phase_channels);
}

int check(int32_t batch, int32_t n_in, int32_t n_out, int32_t phase_channels, const void *indices) {
    if (batch < 1 || n_in < 1 || n_out < 1 || n_out > n_in || (phase_channels != 0 && phase_channels != 1 && phase_channels != 3) ||
        (size_t)batch * (size_t)n_in >= (1ull << 31)) {
        fgs_set_error("fgs_gather: invalid dims B=%d n_in=%d n_out=%d phase_channels=%d", batch, n_in, n_out, phase_channels);
        return FGS_EINVAL;
    }
    if (!indices) { fgs_set_error("fgs_gather: null indices"); return FGS_EINVAL; }
    return FGS_OK;
}

}  // namespace

extern "C" {

int fgs_gather_forward(int32_t batch, int32_t n_in, int32_t n_out, int32_t phase_channels, const int64_t *indices,
                       const float *pos, const float *scale, const float *quat, const float *color,
                       const float *opacity, const float *phase, float *o_pos, float *o_scale, float *o_quat,
                       float *o_color, float *o_opacity, float *o_phase, void *stream) {
    int rc = check(batch, n_in, n_out, phase_channels, indices);
    if (rc) return rc;
    if (!pos || !scale || !quat || !color || !opacity || !o_pos || !o_scale || !o_quat || !o_color || !o_opacity ||
        (phase_channels && (!phase || !o_phase))) {
        fgs_set_error("fgs_gather_forward: null pointer argument");
        return FGS_EINVAL;
    }
    const int32_t total = batch * n_out;
    GatherPtrs p{pos, scale, quat, color, opacity, phase_channels ? phase : nullptr,
                 o_pos, o_scale, o_quat, o_color, o_opacity, phase_channels ? o_phase : nullptr};
    hipLaunchKernelGGL(k_gather<false>, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       total, n_in, n_out, phase_channels, reinterpret_cast<const long long *>(indices), p);
    FGS_LAUNCH_CHECK("k_gather");
    return FGS_OK;
}

int fgs_gather_backward(int32_t batch, int32_t n_in, int32_t n_out, int32_t phase_channels, const int64_t *indices,
                        const float *g_o_pos, const float *g_o_scale, const float *g_o_quat, const float *g_o_color,
                        const float *g_o_opacity, const float *g_o_phase, float *g_pos, float *g_scale, float *g_quat,
                        float *g_color, float *g_opacity, float *g_phase, void *stream) {
    int rc = check(batch, n_in, n_out, phase_channels, indices);
    if (rc) return rc;
    if (!g_o_pos || !g_o_scale || !g_o_quat || !g_o_color || !g_o_opacity || !g_pos || !g_scale || !g_quat || !g_color ||
        !g_opacity || (phase_channels && (!g_o_phase || !g_phase))) {
        fgs_set_error("fgs_gather_backward: null pointer argument");
        return FGS_EINVAL;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t bn = (size_t)batch * n_in;
    hipError_t e = hipMemsetAsync(g_pos, 0, bn * 3 * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(g_scale, 0, bn * 3 * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(g_quat, 0, bn * 4 * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(g_color, 0, bn * 3 * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(g_opacity, 0, bn * 4, st);
    if (e == hipSuccess && phase_channels) e = hipMemsetAsync(g_phase, 0, bn * phase_channels * 4, st);
    if (e != hipSuccess) { fgs_set_error("fgs_gather_backward memset: %s", hipGetErrorString(e)); return FGS_ELAUNCH; }
    const int32_t total = batch * n_out;
    GatherPtrs p{g_pos, g_scale, g_quat, g_color, g_opacity, phase_channels ? g_phase : nullptr,
                 const_cast<float *>(g_o_pos), const_cast<float *>(g_o_scale), const_cast<float *>(g_o_quat),
                 const_cast<float *>(g_o_color), const_cast<float *>(g_o_opacity),
                 phase_channels ? const_cast<float *>(g_o_phase) : nullptr};
    hipLaunchKernelGGL(k_gather<true>, dim3((total + 255) / 256), dim3(256), 0, st, total, n_in, n_out, phase_channels,
                       reinterpret_cast<const long long *>(indices), p);
    FGS_LAUNCH_CHECK("k_scatter");
    return FGS_OK;
}

}  // extern "C"
