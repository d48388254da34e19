// hipFFT (rocFFT) plan cache shared by the angular-spectrum renderer, the standalone propagator and the spectral
// losses: batched C2C transforms (2-D, and 1-D rows), in place, caller-provided work areas; and fgs_fft2_exec, the 2-D
// transform with our own column pass (end of file).
#include <hipfft/hipfft.h>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include "fgs_internal.h"
#include "fgs_colfft.h"

namespace {

// ---- hipFFT plan cache -----------------------------------------------------------------------
// One plan per (device, H, W, batch), built on first use (host work; later calls only enqueue).  A hipFFT handle
// carries its stream and work area as mutable state, so every use of a handle -- SetStream, SetWorkArea, Exec --
// happens under that plan's own mutex: callers on different streams or threads that render the same shape on the
// same device are serialised on the HOST for the few microseconds of the enqueue, and each transform runs on the
// stream and in the work area of the call that enqueued it.  Plans are never shared between devices.
struct FftKey {
    int dev, h, w, batch;
    bool operator<(const FftKey &o) const { return std::tie(dev, h, w, batch) < std::tie(o.dev, o.h, o.w, o.batch); }
};
struct FftPlan { hipfftHandle handle = 0; size_t work = 0; std::mutex mu; };
std::mutex g_fft_mu;  // guards the map only
std::map<FftKey, std::unique_ptr<FftPlan>> g_fft;

int get_fft_plan(int H, int W, int batch, FftPlan **out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { fgs_set_error("hipGetDevice failed"); return FGS_ELAUNCH; }
    std::lock_guard<std::mutex> lk(g_fft_mu);
    const FftKey key{dev, H, W, batch};
    auto it = g_fft.find(key);
    if (it != g_fft.end()) { *out = it->second.get(); return FGS_OK; }
    std::unique_ptr<FftPlan> pl(new FftPlan());
    int n[2] = {H, W};
    if (hipfftCreate(&pl->handle) != HIPFFT_SUCCESS) { fgs_set_error("hipfftCreate failed"); return FGS_ELAUNCH; }
    hipfftResult r = hipfftSetAutoAllocation(pl->handle, 0);
    if (r == HIPFFT_SUCCESS) {
        if (H > 0)
            r = hipfftMakePlanMany(pl->handle, 2, n, nullptr, 1, H * W, nullptr, 1, H * W, HIPFFT_C2C, batch, &pl->work);
        else  // H == 0: `batch` contiguous rows of length W, 1-D transforms (the row pass of a 2-D transform)
            r = hipfftMakePlanMany(pl->handle, 1, n + 1, nullptr, 1, W, nullptr, 1, W, HIPFFT_C2C, batch, &pl->work);
    }
    if (r != HIPFFT_SUCCESS) {
        (void)hipfftDestroy(pl->handle);
        fgs_set_error("hipfftMakePlanMany(%dx%d x%d) failed: %d", H, W, batch, (int)r);
        return FGS_ELAUNCH;
    }
    *out = pl.get();
    g_fft[key] = std::move(pl);
    return FGS_OK;
}

}  // namespace

int fgs_fft_exec(int H, int W, int batch, float2 *data, int dir, void *work, hipStream_t st) {
    FftPlan *pl = nullptr;
    const int rc = get_fft_plan(H, W, batch, &pl);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(pl->mu);
    if (hipfftSetStream(pl->handle, st) != HIPFFT_SUCCESS || hipfftSetWorkArea(pl->handle, work) != HIPFFT_SUCCESS) {
        fgs_set_error("hipfft stream/work-area setup failed");
        return FGS_ELAUNCH;
    }
    const hipfftResult r = hipfftExecC2C(pl->handle, reinterpret_cast<hipfftComplex *>(data),
                                         reinterpret_cast<hipfftComplex *>(data), dir);
    if (r != HIPFFT_SUCCESS) { fgs_set_error("hipfftExecC2C failed: %d", (int)r); return FGS_ELAUNCH; }
    return FGS_OK;
}


int fgs_fft_work_bytes(int H, int W, int batch, size_t *bytes) {
    FftPlan *pl = nullptr;
    const int rc = get_fft_plan(H, W, batch, &pl);
    if (rc) return rc;
    *bytes = pl->work;
    return FGS_OK;
}

// 1-D C2C transforms of `rows` contiguous rows of length W, in place (the row pass of the column-fused 2-D transforms
// of the angular-spectrum renderer; plans cached like the 2-D ones, key H = 0).
int fgs_fft_rows_exec(int W, int rows, float2 *data, int dir, void *work, hipStream_t st) {
    return fgs_fft_exec(0, W, rows, data, dir, work, st);
}
int fgs_fft_rows_work_bytes(int W, int rows, size_t *bytes) { return fgs_fft_work_bytes(0, W, rows, bytes); }

// ---- 2-D transforms with our own column pass ---------------------------------------------------------------------
// rocFFT's 2-D C2C plans run a row kernel at ~5.4 TB/s and a column kernel at ~2 TB/s (config 5's 805 MB: 0.30 + 0.80 ms).
// For heights 64 ... 1024 that are powers of two fgs_fft2_exec uses rocFFT for the rows only and k_colfft_plain for the
// columns: per (field, tile of TC columns) the H x TC tile goes through LDS (128-byte row segments), radix-8 FFT down the
// columns (fgs_colfft.h), one read and one write of the data.  Other heights: rocFFT's 2-D plan.  Unnormalised both ways.
namespace {

// AMAX (inverse only): the block also leaves the maximum of sqrt(|u / (H W)|^2 + 1e-8) over its tile in
// amax[(field / 3) * slots + (field % 3) * gridDim.x + blockIdx.x] -- the per-image maximum of the ASM renderer's amplitude
// (DR:1316-1322) as block partials, for free in the pass that produces the values (k_asm_max was a launch of its own: 10 us at one
// config-5 image); the value is fgs_asm_amplitude's, as everywhere.  Slots the tiles do not fill are zeroed by block (0, 3 b).
template <int LOGN, int TC, bool INV, bool AMAX = false>
__global__ __launch_bounds__((1 << LOGN) * TC / 8) void k_colfft_plain(int W, float2 *__restrict__ data, float *__restrict__ amax = nullptr,
                                                                       int slots = 0, float inv_hw = 0.0f) {
    constexpr int N = 1 << LOGN, PER = 8, NT = N * TC / PER, E1 = N / 8;
    constexpr bool INNER = lds_fft_inner_in_registers<LOGN>();
    __shared__ float2 x[N][TC];
    __shared__ float2 tw[N / 2];
    for (int n = threadIdx.x; n < N / 2; n += NT) {
        float sn, cs;
        sincospif(-2.0f * (float)n / (float)N, &sn, &cs);
        tw[n] = make_float2(cs, sn);
    }
    const int c0 = blockIdx.x * TC, col = threadIdx.x % TC, q = threadIdx.x / TC;
    const bool live = c0 + col < W;
    float2 *f = data + (size_t)blockIdx.y * N * W + c0 + col;
    // As in the column kernels of the angular-spectrum renderer (fgs_asm.hip): the block-size-N pass runs in registers on the thread's
    // own eight elements (rows q + e N/8: first forward, straight from HBM; last inverse, straight to HBM) and, for N = 8^k, so does
    // the block-size-8 pass (rows 8 q + e, unit twiddles).  LDS row r holds frequency bitrev(r) between the two.
    float2 v[PER];
    if (!INV) {
#pragma unroll
        for (int e = 0; e < PER; ++e) v[e] = live ? f[(size_t)(q + e * E1) * W] : make_float2(0.0f, 0.0f);
        __syncthreads();  // twiddles
        oct_dif<false>(v, tw[q], tw[2 * q], tw[4 * q]);
#pragma unroll
        for (int e = 0; e < PER; ++e) x[q + e * E1][col] = v[e];
        __syncthreads();
        lds_fft_columns<LOGN, TC, NT, false, true, INNER>(x, tw);
#pragma unroll
        for (int e = 0; e < PER; ++e) v[e] = x[INNER ? 8 * q + e : q + e * E1][col];
        if (INNER) oct_dif<true>(v, v[0], v[0], v[0]);
        if (live) {
#pragma unroll
            for (int e = 0; e < PER; ++e) f[(size_t)bitrev<LOGN>(INNER ? 8 * q + e : q + e * E1) * W] = v[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < PER; ++e)
            v[e] = live ? f[(size_t)bitrev<LOGN>(INNER ? 8 * q + e : q + e * E1) * W] : make_float2(0.0f, 0.0f);
        if (INNER) oct_dit<true>(v, v[0], v[0], v[0]);
#pragma unroll
        for (int e = 0; e < PER; ++e) x[INNER ? 8 * q + e : q + e * E1][col] = v[e];
        __syncthreads();  // (twiddles too)
        lds_fft_columns<LOGN, TC, NT, true, true, INNER>(x, tw);
#pragma unroll
        for (int e = 0; e < PER; ++e) v[e] = x[q + e * E1][col];
        oct_dit<false>(v, tw[q], tw[2 * q], tw[4 * q]);
        if (live) {
#pragma unroll
            for (int e = 0; e < PER; ++e) f[(size_t)(q + e * E1) * W] = v[e];
        }
        if constexpr (AMAX) {
            float mx = 0.0f;
            if (live) {
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    mx = fmaxf(mx, fgs_asm_amplitude(fgs_asm_intensity(v[e], inv_hw)));  // (the value the backward compares with)
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            __shared__ float wmx[NT / 64 > 0 ? NT / 64 : 1];
            if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = mx;
            __syncthreads();
            const int field = blockIdx.y, img = field / 3, ch = field - 3 * img;
            float *dst = amax + (size_t)img * slots;
            if (threadIdx.x == 0) {
                float m = wmx[0];
                for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, wmx[w]);
                dst[ch * gridDim.x + blockIdx.x] = m;
            }
            if (ch == 0 && blockIdx.x == 0)
                for (int i = 3 * (int)gridDim.x + threadIdx.x; i < slots; i += NT) dst[i] = 0.0f;
        }
    }
}

int colfft_logn(int H) {
    for (int lg = 6; lg <= 10; ++lg)
        if (H == (1 << lg)) return lg;
    return 0;
}

template <bool INV, bool AMAX = false>
int launch_colfft_plain(int lg, int W, int batch, float2 *data, hipStream_t st, float *amax = nullptr, int slots = 0,
                        float inv_hw = 0.0f) {
#define FGS_CP(LG, TCV)                                                                                              \
    hipLaunchKernelGGL((k_colfft_plain<LG, TCV, INV, AMAX>), dim3((unsigned)((W + TCV - 1) / TCV), (unsigned)batch),   \
                       dim3((1 << LG) * TCV / 8), 0, st, W, data, amax, slots, inv_hw)
    switch (lg) {
        case 6: FGS_CP(6, 16); break;
        case 7: FGS_CP(7, 16); break;
        case 8: FGS_CP(8, 16); break;
        case 9: FGS_CP(9, 16); break;
        default: FGS_CP(10, 8); break;
    }
#undef FGS_CP
    FGS_LAUNCH_CHECK("k_colfft_plain");
    return FGS_OK;
}

}  // namespace

int fgs_fft2_work_bytes(int H, int W, int batch, size_t *bytes) {
    if (colfft_logn(H)) return fgs_fft_rows_work_bytes(W, batch * H, bytes);
    return fgs_fft_work_bytes(H, W, batch, bytes);
}

int fgs_fft2_exec(int H, int W, int batch, float2 *data, int dir, void *work, hipStream_t st) {
    const int lg = colfft_logn(H);
    if (!lg) return fgs_fft_exec(H, W, batch, data, dir, work, st);
    int rc = fgs_fft_rows_exec(W, batch * H, data, dir, work, st);
    if (rc) return rc;
    return dir == HIPFFT_FORWARD ? launch_colfft_plain<false>(lg, W, batch, data, st)
                                 : launch_colfft_plain<true>(lg, W, batch, data, st);
}

// Inverse 2-D transform of `images` x 3 fields that also leaves every image's amplitude maximum as `slots` block partials in
// amax[image][slots] (see k_colfft_plain<..., AMAX>).  *fused = false (nothing written to amax) when the frame takes rocFFT's
// 2-D plan or its column tiles do not fit the slots: the caller then runs its own reduction.
int fgs_fft2_inverse_with_max(int H, int W, int images, float2 *data, void *work, float *amax, int slots, float inv_hw,
                              bool *fused, hipStream_t st) {
    const int lg = colfft_logn(H);
    const int tiles = (W + (lg == 10 ? 8 : 16) - 1) / (lg == 10 ? 8 : 16);
    *fused = lg != 0 && 3 * tiles <= slots;
    if (!*fused) return fgs_fft2_exec(H, W, images * 3, data, HIPFFT_BACKWARD, work, st);
    const int rc = fgs_fft_rows_exec(W, images * 3 * H, data, HIPFFT_BACKWARD, work, st);
    if (rc) return rc;
    return launch_colfft_plain<true, true>(lg, W, images * 3, data, st, amax, slots, inv_hw);
}
