// Internal declarations shared by the HIP translation units of libfgs_hip.so.
// gfx950 (MI355X, CDNA4) only: wave64, 160 KiB LDS/CU, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fgs.h"

#include "fgs_plan.h"

#define FGS_LAUNCH_CHECK(what)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            fgs_set_error("%s: %s", what, hipGetErrorString(e__));          \
            return FGS_ELAUNCH;                                             \
        }                                                                   \
    } while (0)

enum FgsStage { ST_PROJECT = 0, ST_DEPTH_SORT, ST_DUP_EMIT, ST_TILE_SORT, ST_TILE_RANGES, ST_COMPOSITE_FWD,
                ST_COMPOSITE_BWD, ST_PROJECT_BWD, ST_SPLAT_FWD, ST_FIELD_FWD, ST_FIELD_BWD, ST_SPLAT_BWD };
static_assert(ST_SPLAT_BWD + 1 == FGS_NUM_STAGES, "stage list and FGS_NUM_STAGES disagree");
void fgs_stage_begin(int stage, hipStream_t st);  // no-ops unless fgs_stage_timing_enable(1)
void fgs_stage_end(int stage, hipStream_t st);

// ---- stage launchers (each enqueues on `st`, returns FGS_OK / FGS_ELAUNCH) ----
// plane_* != 0 selects the ASM variant: additionally writes saved.layer = nearest depth plane
// (DR:1136-1148) of each Gaussian
int fgs_launch_project(const FgsPlan &p, const float *cams, const float *pos, const float *scale,
                       const float *quat, const float *color, const float *opacity, char *saved,
                       hipStream_t st, int num_planes = 0, float plane_near = 0.0f, float plane_far = 0.0f,
                       uint32_t *zero_words = nullptr /* cleared on the side: the depth sort's hand-off words */, uint32_t zero_count = 0);
int fgs_launch_project_bwd(const FgsPlan &p, const float *cams, const float *pos, const float *scale,
                           const float *quat, const char *saved, const float *grad_rows, float *g_pos,
                           float *g_scale, float *g_quat, float *g_color, float *g_opacity, float *g_phase,
                           hipStream_t st, float *row_sums = nullptr /* scratch [B*N][12]: blend path row totals */);

int fgs_launch_asm_project_bwd(const FgsPlan &p, const float *cams, const float *pos, const float *scale,
                               const float *quat, const float *color, const float *phase, int phase_channels,
                               const char *saved, const float *grad_rows, float *g_pos, float *g_scale,
                               float *g_quat, float *g_color, float *g_opacity, float *g_phase, hipStream_t st,
                               bool wave_rows = false);

// Stable LSD radix sort of (key,val) uint32 pairs over `num_segs` independent segments.
// Segment s covers elements [s*seg_stride, s*seg_stride + len) with len = seg_len (host) or
// *seg_len_dev (device, single segment).  Sorts bits [0, key_bits).  The sorted result is
// left in (keys_out, vals_out); in/alt buffers are clobbered.
int fgs_launch_radix_sort(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_alt, uint32_t *vals_alt,
                          uint32_t *vals_final /*nullable: where the last pass writes vals*/,
                          uint32_t **keys_sorted /*out: which buffer holds sorted keys*/,
                          uint32_t **vals_sorted, uint32_t seg_len, const uint32_t *seg_len_dev,
                          uint32_t seg_capacity, uint32_t seg_stride, uint32_t num_segs, uint32_t key_bits,
                          uint32_t *hist, hipStream_t st, const uint32_t *keys_first = nullptr,
                          uint32_t index_payload_mod = 0,
                          const uint32_t *key_stats = nullptr /* depth sort: k_project's per-block OR / AND of the visible keys:
                                                                 sort by the bits that vary, skip the passes nobody needs */,
                          uint32_t key_recs = 0 /* records per segment */,
                          int pass_mode = 0 /* 0 = automatic (fused single-launch passes for segments of <= 4096 keys) | 1 = fused
                                               passes, 11-bit digits, up to 64 K keys | 2 = fused passes, 8-bit digits | 3 = fused
                                               passes, 8-bit digits, per-block histograms HANDED OFF between the blocks through `hist`
                                               (which the caller cleared beforehand: fgs_sort.hip) */);

int fgs_launch_binning(const FgsPlan &p, char *saved, char *scratch, hipStream_t st);

// Batched in-place 2-D C2C FFT of `batch` H x W complex fields (fgs_fft.hip: plans cached per (device, H, W, batch),
// built on first use; `work` = caller's work area of fgs_fft_work_bytes bytes).  dir: HIPFFT_FORWARD (-1) /
// HIPFFT_BACKWARD (+1), both unnormalised.
int fgs_fft_work_bytes(int H, int W, int batch, size_t *bytes);
int fgs_fft_exec(int H, int W, int batch, float2 *data, int dir, void *work, hipStream_t st);
int fgs_fft_rows_exec(int W, int rows, float2 *data, int dir, void *work, hipStream_t st);  // 1-D, along rows of length W
int fgs_fft_rows_work_bytes(int W, int rows, size_t *bytes);
// 2-D transform, rocFFT rows + our own column pass for power-of-two heights 64 ... 1024 (else rocFFT's 2-D plan)
int fgs_fft2_work_bytes(int H, int W, int batch, size_t *bytes);
int fgs_fft2_exec(int H, int W, int batch, float2 *data, int dir, void *work, hipStream_t st);
// inverse transform of images x 3 fields + per-image block maxima of sqrt(|u inv_hw|^2 + 1e-8) (fgs_fft.hip); *fused tells whether amax was written
int fgs_fft2_inverse_with_max(int H, int W, int images, float2 *data, void *work, float *amax, int slots, float inv_hw,
                              bool *fused, hipStream_t st);
int fgs_launch_composite_fwd(const FgsPlan &p, const float *phase, char *saved, float *out_rgb,
                             float *out_depth, hipStream_t st);
int fgs_launch_composite_bwd(const FgsPlan &p, const float *phase, const char *saved, char *scratch,
                             const float *g_rgb, const float *g_depth, float *g_phase, hipStream_t st);
int fgs_launch_count_pairs(const FgsPlan &p, const char *saved, uint64_t *out, hipStream_t st);

// ---- small device helpers ----
__device__ __forceinline__ uint32_t fgs_lane() { return threadIdx.x & 63u; }

// XCD-aware block remap (cdna_hip_programming.md T1).  Workgroups are dealt round-robin over the 8 XCDs, each with
// its own L2, so blocks bid and bid + 1 never share an L2.  remap(bid) gives every XCD a CONTIGUOUS range of
// logical work items: neighbours in the work list (adjacent depth-rank blocks of one image, adjacent tiles) then hit
// the same L2 -- shared records are fetched once and, more importantly, partial-line writes to neighbouring
// addresses (4-byte list entries, 40-byte gradient rows) merge in that L2 instead of reaching HBM as masked
// partial writes from several XCDs.  Bijective on [0, nwg) for any nwg; placement only, never correctness.
__device__ __forceinline__ uint32_t fgs_xcd_remap(uint32_t bid, uint32_t nwg) {
    const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, i = bid >> 3;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + i;
}

// order-preserving map float -> uint32 (ascending), -0.0 folded into +0.0
__device__ __forceinline__ uint32_t fgs_float_key(float f) {
    f = f + 0.0f;
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// A product or sum the compiler must not fuse into a neighbouring operation: the value passes through an empty asm, which makes
// it opaque.  (hipcc's __fmul_rn / __fadd_rn are plain operators -- they do NOT stop FMA contraction -- and a
// `#pragma clang fp contract(off)` does not either in a unit compiled with the default -ffp-contract=fast: the AMDGPU backend
// fuses under the global setting.  Found in round 4: a "separately rounded" expression written with the intrinsics came out
// as v_mul + v_fmac in fgs_fft.hip.)
__device__ __forceinline__ float fgs_rounded(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

// kz^2 = (1/l)^2 - FX^2 - FY^2 exactly as the reference's fp32 expression evaluates it (DR:993): every product and every
// difference rounded on its own, whatever the translation unit's contraction setting.  Near the evanescent boundary the
// difference cancels to a few ulps of 1/l^2 and dkz/dlambda = -1 / (l^3 kz) weights exactly those frequencies most, so the
// forward table, the recurrence factor D, both dL/dlambda kernels and the standalone propagator (fgs_spectral.hip) take kz^2
// from this ONE function: forward and adjoint agree on kz to the bit.
__device__ __forceinline__ float fgs_kz2(float il, float fx, float fy) {
    const float a = fgs_rounded(il * il), b = fgs_rounded(fx * fx), c = fgs_rounded(fy * fy);
    return fgs_rounded(fgs_rounded(a - b) - c);
}

// |U|^2 of one complex sample of the ASM renderer's total field and sqrt(|U|^2 + 1e-8) (DR:1316-1319), every operation rounded on
// its own.  The per-image maximum of the latter is found in one translation unit (the inverse column FFT's epilogue, fgs_fft.hip)
// and COMPARED FOR EQUALITY with recomputed values in another (k_asm_output_bwd*, torch.max's gradient goes to the pixels that
// equal the maximum): producer and consumers take the value from here, so they agree to the bit whatever their units' flags.
__device__ __forceinline__ float fgs_asm_intensity(float2 u, float inv_hw) {
    const float ur = fgs_rounded(u.x * inv_hw), ui = fgs_rounded(u.y * inv_hw);
    return fgs_rounded(fgs_rounded(ur * ur) + fgs_rounded(ui * ui));
}
__device__ __forceinline__ float fgs_asm_amplitude(float intensity) { return sqrtf(fgs_rounded(intensity + 1e-8f)); }
