// Angular-spectrum wave-field renderer (BASELINE config 5): replaces ASMWaveFieldRenderer.forward
// (DR:1150-1344) and AngularSpectrumPropagator (DR:929-1065) and their autograd.
//
// Pipeline (forward), all on the caller's stream:
//   k_project (shared; also assigns the nearest depth plane) -> lists per (image, plane, tile) (layered mask binning,
//   fgs_bin.hip; no depth sort: the sums below are order-independent, one stable pass groups the Gaussians by plane) -> k_asm_prep (phasors c e^{i phi} per Gaussian, transfer functions H = exp(i 2 pi z_p sqrt(max(l_c^-2 -
//   fx^2 - fy^2, 0))) (DR:989-999), FFT twiddles: one launch) -> k_asm_splat (one wave per list: complex amplitudes
//   a c e^{i phi}, order-independent sum, DR:1233-1283) -> 2-D forward transform of all B*P*3 plane fields: rocFFT 1-D rows
//   + k_colfft_fwd, our own column FFT fused with acc_c = sum_p F_pc H_pc (heights 2^6 ... 2^10; otherwise rocFFT's 2-D plan
//   + k_asm_accumulate) -> ONE inverse transform per (image, channel) instead of one per plane (linearity of the
//   propagation, 3 instead of 48) -> k_asm_max / k_asm_output: sqrt(|U|^2 + 1e-8), per-image max-normalisation (two-level
//   reduction, no atomics), background composition and clamps (DR:1315-1332).
// Backward retraces this with the adjoint transforms (k_colfft_bwd: gAcc conj(H) + inverse column FFT, then rocFFT inverse
// rows); dL/dlambda comes from Z_c = sum_p z_p F_pc H_pc, summed by the forward beside acc_c -- no plane spectrum is kept.
//
// Bounds: the splat is VALU-bound on long lists and HBM-bound on config 5's short ones; the transforms are HBM-bound (8 B
// per complex sample per pass, two passes per 2-D transform).
//
// Compiled WITHOUT fast-math: the transfer-function phase reaches ~200 rad, so sin/cos need the
// accurate range reduction of sincosf.
#include <hipfft/hipfft.h>
#include <type_traits>
#include "fgs_internal.h"
#include "fgs_wave.h"
#include "fgs_colfft.h"

namespace {

// an (image, plane) pair -- or, with `tiles` = lists per image, an image -- with no list entry at all: seg_off is the
// exclusive scan of every list's depth-segment count in key order (key = (b P + p) T + t; k_tile_post writes it on both
// list-building paths, [lists + 1] entries), so a key range without entries is a range without units.  (NOT `ranges`: the
// radix path leaves the ranges of empty lists zeroed.)
__device__ __forceinline__ bool asm_plane_empty(const uint32_t *__restrict__ seg_off, uint32_t bp, uint32_t tiles) {
    return seg_off[(size_t)(bp + 1u) * tiles] == seg_off[(size_t)bp * tiles];
}

constexpr float NEG_HALF_LOG2E = -0.72134752044448170368f;
constexpr int ACH = 64;
constexpr int ASM_FWD_PARTS = 4;  // list parts (waves) per (image, plane, tile) in the forward splat
constexpr uint32_t ASM_ONE_WAVE_LISTS = 24576;  // from this many lists per launch on: one wave per list in the forward splat (measured: 16 384 lists 0.105 ms with four parts vs 0.139 with one, 131 072 lists 0.83 vs 0.345)
constexpr int RED_BLOCKS = 128;   // blocks (= partials) per image of the per-image scalar reductions, see below

struct AsmPlan {
    FgsAsmDims a;
    FgsPlan base;        // projection + binning with layers = num_planes
    size_t HW;
    // saved sections (after base.L.total_bytes)
    size_t v_field;      // float2 [B][P][3][H][W]  plane fields -> spectra (kept for the backward)
    size_t v_htab;       // float2 [3][P][H][W]     transfer functions H_pc, then [3][H][W] their plane-to-plane factor D_c (asm_transfer_block)
    size_t v_total;      // float2 [B][3][H][W]     total field U (unnormalised inverse FFT)
    size_t v_scal;       // float  [B]              per-image maxval
    size_t v_ccs;        // float  [B][N][8]        phasors c cos(phi), c sin(phi) per channel (k_asm_phasors)
    size_t v_tw;         // float2 [H/2]            twiddles of the column-fused transforms
    size_t v_zsum;       // float2 [B][3][H][W]     Z_c = sum_p z_p F_pc H_pc (column-fused path): all the backward needs of the
                         //                         spectra for dL/dlambda, so the spectra themselves are never stored
    int col_logn;        // log2(H) when the column direction runs in k_colfft_* (H = 64 ... 1024, a power of two), else 0
    int col_tc, col_pg;  // its column tile width, and plane groups per image (> 1 for launches that would not fill the chip)
    size_t c_accp;       // float2 [B][col_pg][3][H][W] partial plane sums (col_pg > 1)
    size_t v_total_bytes;
    // scratch sections (after base.s_total)
    size_t c_acc;        // float2 [B][3][H][W]
    size_t c_part;       // float2 [B][RED_BLOCKS] block partials of the per-image scalars | double [3][B*HW/256] of dL/dlambda
    size_t c_fftwork;
    size_t c_total_bytes;
    size_t work_big, work_small;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int make_asm_plan(const FgsAsmDims *a, AsmPlan *p, bool need_fft) {
    if (!a) { fgs_set_error("null dims"); return FGS_EINVAL; }
    if (a->num_planes < 1 || a->num_planes > 64 || (a->phase_channels != 1 && a->phase_channels != 3) ||
        !(a->pixel_pitch > 0.0)) {
        fgs_set_error("invalid ASM dims: planes=%d phase_channels=%d pitch=%g", a->num_planes, a->phase_channels,
                      (double)a->pixel_pitch);
        return FGS_EINVAL;
    }
    FgsDims d{};
    d.batch = a->batch; d.num_gaussians = a->num_gaussians; d.width = a->width; d.height = a->height;
    d.max_radius = a->max_radius;
    for (int i = 0; i < 3; ++i) d.background[i] = a->background[i];
    d.use_phase = 0; d.phase_amplitude = 0.0f; d.num_cameras = a->num_cameras;
    d.bin_mode = a->bin_mode;
    p->a = *a;
    const int rc = fgs_make_plan(&d, &p->base, a->num_planes, false);
    if (rc) return rc;
    const size_t B = a->batch, P = a->num_planes, HW = (size_t)a->width * a->height;
    p->HW = HW;
    // (The forward splat keeps its longest-lists-first order over the whole launch.  Grouping it by image, last image first, so that
    // rocFFT's ascending row pass finds the planes it reads first in the memory-side cache, was measured at config 5, 8 images: row pass
    // 300 -> 261 us, k_colfft_fwd -4 us, but the splat itself 249 -> 312 us and k_tile_post 9 -> 22 us: net +1 %.)
    size_t o = p->base.L.total_bytes;
    p->v_field = o; o = align256(o + B * P * 3 * HW * 8);
    p->v_htab = o; o = align256(o + 3 * (P + 1) * HW * 8);
    p->v_total = o; o = align256(o + B * 3 * HW * 8);
    p->v_scal = o; o = align256(o + B * 4 * 4);
    p->v_ccs = o; o = align256(o + B * (size_t)a->num_gaussians * 8 * 4);
    p->v_tw = o; o = align256(o + 512 * 8);
    p->v_zsum = o; o = align256(o + B * 3 * HW * 8);
    p->col_logn = 0;
    // (whole column tiles only: widths that are multiples of the tile's 16 -- 8 for H = 1024 -- columns; the kernels carry no
    // per-lane guards, so that every load of their plane loops is unconditional and the compiler's wait counts stay exact)
    for (int lg = 6; lg <= 10; ++lg)
        if (a->height == (1 << lg) && a->width % (lg == 10 ? 8 : 16) == 0) p->col_logn = lg;
#ifdef FGS_NO_COLFFT  // experiment builds: rocFFT's 2-D plans for every frame
    p->col_logn = 0;
#endif
#ifndef FGS_COLFFT_TC9
#define FGS_COLFFT_TC9 16  // column tile width of the 512-row kernels (experiment builds: 8)
#endif
    p->col_tc = p->col_logn == 10 ? 8 : p->col_logn == 9 ? FGS_COLFFT_TC9 : 16;
    p->col_pg = 1;
    if (p->col_logn) {
        const size_t blocks = (size_t)((a->width + p->col_tc - 1) / p->col_tc) * 3 * B;
        // (at least 512 blocks.  Choosing PG by "rounds of 256 blocks x planes per block" instead -- 8 groups of 2 planes for one
        // config-5 image rather than 6 of 3 -- was measured: 0.491 -> 0.518 ms at one image, 0.714 -> 0.733 at two; the blocks do not
        // run in rounds, and every group costs a partial acc / Z pair, a table plane and a block start-up)
        if (blocks < 512) p->col_pg = (int)((512 + blocks - 1) / blocks);
        if (p->col_pg > (int)P) p->col_pg = (int)P;
    }
    // (Round 3 built the row direction fused with the splat / its adjoint -- one kernel per direction keeping a band of 8 rows
    // x 3 channels x 512 points in LDS, the plane fields never in HBM -- and measured it SLOWER: config 5 at 8 images 2.63 ->
    // 3.33 ms.  133 KB of LDS = one 16-wave block per CU, whose phases (latency-bound list walk, then the transforms) run back
    // to back with nothing to overlap them.  Removed in round 4; DESIGN_LOG.md 10.5, profiles/r03_ab_config5_*_rows_fused.txt,
    // the code is in the history at 763bed2 csrc/fgs_asm_rows.h.)
    p->v_total_bytes = o;
    p->work_big = p->work_small = 0;
    if (need_fft) {
        int r2 = p->col_logn ? fgs_fft_rows_work_bytes(a->width, (int)(B * P * 3) * a->height, &p->work_big)
                             : fgs_fft_work_bytes(a->height, a->width, (int)(B * P * 3), &p->work_big);
        if (r2) return r2;
        r2 = fgs_fft2_work_bytes(a->height, a->width, (int)(B * 3), &p->work_small);
        if (r2) return r2;
    }
    o = p->base.s_total;
    p->c_acc = o; o = align256(o + B * 3 * HW * 8);
    {
        size_t nwl = (B * HW + 255) / 256;  // dL/dlambda partials: per block of k_asm_accumulate_bwd / k_colfft_bwd
        const size_t nwl2 = (size_t)((a->width + p->col_tc - 1) / p->col_tc) * B * p->col_pg;
        if (nwl2 > nwl) nwl = nwl2;
        p->c_part = o; o = align256(o + B * RED_BLOCKS * 8 + 3 * nwl * 8);  // (the dL/dlambda partials are doubles)
    }
    p->c_accp = o;
    if (p->col_pg > 1) o = align256(o + 2 * B * p->col_pg * 3 * HW * 8);  // partial plane sums of acc and of Z
    p->c_fftwork = o; o = align256(o + (p->work_big > p->work_small ? p->work_big : p->work_small) + 256);
    p->c_total_bytes = o;
    return FGS_OK;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// fftfreq(n, d)[k] as torch computes it: integer index (negative upper half) times 1/(n d)
__device__ __forceinline__ float fftfreq(int k, int n, float inv_nd) {
    const int ks = (k < (n + 1) / 2) ? k : k - n;
    return (float)ks * inv_nd;
}

// depth planes: torch.linspace(near, far, P), DR:1106
__device__ __forceinline__ float plane_depth(int k, int P, float near_, float far_) {
    if (P == 1) return near_;  // torch.linspace(near, far, 1) = [near]
    const float step = (far_ - near_) / (float)(P - 1);
    return (k < P / 2) ? near_ + step * (float)k : far_ - step * (float)(P - 1 - k);
}

// H[c][p][ky][kx] = exp(i * ((2 pi * z_p) * kz)), kz = sqrt(max(1/l_c^2 - fx^2 - fy^2, 0))   DR:989-999
// fftfreq(n - k) = -fftfreq(k) exactly, and H depends on fx^2, fy^2 only: one thread evaluates the quadrant entry
// (ky <= H/2, kx <= W/2) -- the accurate sincosf is what this kernel costs -- and stores it at its up to four mirror
// positions, bit-identical to evaluating every entry (58 -> 20 us for 3 x 16 planes of 512^2).
// Behind the 3 P planes: D[c][ky][kx] = exp(i * ((2 pi * -step) * kz)), step = (far - near) / (P - 1) -- the planes are equally spaced
// (torch.linspace, DR:1194), so H_(p+1) = H_p D and the column kernels walk the planes by this recurrence instead of reading H_p
// (k_colfft_fwd / k_colfft_bwd; the table itself is still what k_asm_accumulate[_bwd] and each block's first plane read).
// `pstride` > 1 (the column kernels' path): only the planes 0, pstride, 2 pstride, ... -- the first plane of each plane group, which is
// all that path reads of H -- and D are evaluated (3 x 2 planes instead of 3 x 17 at config 5's 8 images: 22 -> 5 us per forward).
__device__ __forceinline__ void asm_transfer_block(uint32_t blk, int W, int H, int P, int pstride, float near_, float far_, float focal,
                                                   float inv_ndx, float inv_ndy,
                                                   const float *__restrict__ wavelengths,
                                                   float2 *__restrict__ htab) {
    const size_t HW = (size_t)W * H;
    const int QW = W / 2 + 1, QH = H / 2 + 1;
    const size_t QHW = (size_t)QW * QH;
    const size_t i = (size_t)blk * 256 + threadIdx.x;
    const int np = (P + pstride - 1) / pstride;  // planes evaluated per channel (+ D)
    if (i >= 3 * (size_t)(np + 1) * QHW) return;
    const int kx = (int)(i % QW), ky = (int)((i / QW) % QH);
    const int pi = (int)((i / QHW) % (np + 1)), c = (int)(i / (QHW * (np + 1)));
    const int p = pi < np ? pi * pstride : P;
    const float fx = fftfreq(kx, W, inv_ndx), fy = fftfreq(ky, H, inv_ndy);
    const float il = 1.0f / wavelengths[c];
    float kz2 = fgs_kz2(il, fx, fy);
    kz2 = kz2 < 0.0f ? 0.0f : kz2;
    const float kz = sqrtf(kz2);
    float sn, cs;
    if (p < P) {  // H_p: the reference's own fp32 expression, rounding for rounding (DR:989-999)
        const float z = focal - plane_depth(p, P, near_, far_);
        const float theta = (6.28318530717958647692f * z) * kz;
        sincosf(theta, &sn, &cs);
    } else {
        // D: its phase is added up to P - 1 times, so it is evaluated in double and only its two components are rounded: 4e-8 rad per
        // step instead of the 1.5e-6 (up to 6.6e-6) of the fp32 expression -- H_lo D^k then stays within the reference's own fp32 phase
        // rounding (5e-6 rad typical at 200 rad) of the directly evaluated H_p for every plane count the interface admits
        const double zd = P > 1 ? -((double)far_ - (double)near_) / (double)(P - 1) : 0.0;
        // (the SAME fp32 kz^2 as H: an exactly evaluated kz^2 differs from the rounded one by up to an ulp of 1/l^2, which near the
        // evanescent boundary is a relative 1e-4 of kz -- H_lo D^k would drift from H_p by that much of its phase)
        const double th = 6.283185307179586476925 * zd * sqrt((double)kz2);
        double sd, cd;
        sincos(th, &sd, &cd);
        sn = (float)sd; cs = (float)cd;
    }
    const float2 h = make_float2(cs, sn);
    float2 *plane = p < P ? htab + ((size_t)c * P + p) * HW : htab + ((size_t)3 * P + c) * HW;
    const int mx = (kx > 0 && W - kx != kx) ? W - kx : -1, my = (ky > 0 && H - ky != ky) ? H - ky : -1;
    plane[(size_t)ky * W + kx] = h;
    if (mx >= 0) plane[(size_t)ky * W + mx] = h;
    if (my >= 0) {
        plane[(size_t)my * W + kx] = h;
        if (mx >= 0) plane[(size_t)my * W + mx] = h;
    }
}

// ccs[g] = (c_r cos phi_r, c_g cos phi_g, c_b cos phi_b, c_r sin phi_r | c_g sin phi_g, c_b sin phi_b, 0, 0)  DR:1274-1283
__device__ __forceinline__ void asm_phasors_block(uint32_t blk, uint32_t total, int phase_channels,
                                                  const float *__restrict__ color, const float *__restrict__ phase,
                                                  float *__restrict__ ccs) {
    const uint32_t g = blk * 256 + threadIdx.x;
    if (g >= total) return;
    float cc[3], cs[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ph = phase_channels == 3 ? phase[3 * (size_t)g + c] : phase[g];
        float sn, co;
        sincosf(ph, &sn, &co);
        const float col = color[3 * (size_t)g + c];
        cc[c] = col * co; cs[c] = col * sn;
    }
    float4 *o = reinterpret_cast<float4 *>(ccs + (size_t)g * 8);
    o[0] = make_float4(cc[0], cc[1], cc[2], cs[0]);
    o[1] = make_float4(cs[1], cs[2], 0.0f, 0.0f);
}

__global__ __launch_bounds__(256) void k_asm_phasors(uint32_t total, int phase_channels, const float *__restrict__ color,
                                                     const float *__restrict__ phase, float *__restrict__ ccs) {
    asm_phasors_block(blockIdx.x, total, phase_channels, color, phase, ccs);
}

// Everything of an ASM forward that depends on the inputs alone, in ONE launch (a one-image step is 45 launches of a few
// microseconds each): blocks [0, nb_ph) the phasors, [nb_ph, nb_ph + nb_tr) the transfer functions, the last block the
// twiddles of the column-fused transforms (N = 0: none).
__global__ __launch_bounds__(256) void k_asm_prep(uint32_t nb_ph, uint32_t nb_tr, uint32_t total, int phase_channels,
                                                  const float *__restrict__ color, const float *__restrict__ phase,
                                                  float *__restrict__ ccs, int W, int H, int P, float near_, float far_,
                                                  float focal, float inv_ndx, float inv_ndy,
                                                  const float *__restrict__ wavelengths, float2 *__restrict__ htab, int N,
                                                  float2 *__restrict__ tw, int pstride) {
    if (blockIdx.x < nb_ph) {
        asm_phasors_block(blockIdx.x, total, phase_channels, color, phase, ccs);
    } else if (blockIdx.x < nb_ph + nb_tr) {
        asm_transfer_block(blockIdx.x - nb_ph, W, H, P, pstride, near_, far_, focal, inv_ndx, inv_ndy, wavelengths, htab);
    } else {
        for (int n = threadIdx.x; n < N / 2; n += 256) {
            float sn, cs;
            sincospif(-2.0f * (float)n / (float)N, &sn, &cs);
            tw[n] = make_float2(cs, sn);
        }
    }
}

// One wave per (image, plane, tile); lane = one pixel of each of the four 8x8 sub-tiles.
// BWD = false: accumulate field += a * c * (cos phi, sin phi) with a = exp(-m/2) * opacity (DR:1263-1283).
// BWD = true : read the field gradient and reduce the twelve per-Gaussian sums into a gradient row.
// WAVE = true (WaveFieldRenderer, DR:832-891): a single layer, and additionally the amplitude-weighted
// depth sums (sum a*depth, sum a) in `dw`; gradient rows are 16 floats wide (slot 12 = dL/ddepth).
// NP = waves per block of the forward (list parts), 1 for the backward.  `ccs` = the Gaussians' phasors c cos(phi),
// c sin(phi) per channel ([B*N][8] floats, k_asm_phasors): the accurate sincosf runs once per Gaussian instead of three
// times per (tile, plane) duplicate at staging time, forward and backward.
template <bool BWD, bool WAVE, int NP>
__global__ __launch_bounds__(64 * NP) void k_asm_splat(
    uint32_t tiles, uint32_t tiles_x, uint32_t P, uint32_t W, uint32_t H, uint32_t dcap,
    const uint32_t *__restrict__ tile_order, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ dup_ids, const float *__restrict__ rec, const float *__restrict__ ccs,
    const uint32_t *__restrict__ dup_off, float2 *__restrict__ field, float *__restrict__ grad_rows,
    float2 *__restrict__ dw, const uint32_t *__restrict__ counters, const uint32_t *__restrict__ seg_off,
    const uint32_t *__restrict__ seg_tile, uint32_t seg_len) {
    // Forward: the splat is a plain sum, so the list is cut into NP parts, one per wave (own LDS staging, no block
    // barrier in the walk) and the partial fields are added in part order at the end -- a launch of few, long lists is
    // latency-bound by the longest (NP = 4); a launch with enough lists to fill the chip (the ASM renderer's (image,
    // plane, tile) lists of a few dozen entries) runs one wave per list with 4 KB of LDS instead of 35 (NP = 1).
    // Backward: one wave per depth-segment unit.
    static_assert(!BWD || NP == 1, "the backward is one wave per unit");
    __shared__ float4 sh0[NP * ACH], sh1[NP * ACH], sh2[NP * ACH], sh3[NP * ACH];
    __shared__ uint32_t shm[NP * ACH], she[BWD ? ACH : 1];
    __shared__ float part[NP == 1 ? 1 : (NP - 1) * (WAVE ? 32 : 24) * 64];
    __shared__ __attribute__((aligned(16))) float red[BWD ? 13 * FGS_RED_PITCH : 4];  // wave_sum_addtid scratch (backward)
    // Forward: one block per (image, plane, tile), longest lists first.  Backward: the splat carries no state
    // along a list, so the work unit is a depth segment of FGS_SEG list entries (unit list of k_tile_order;
    // the grid is sized from the capacity, surplus blocks leave at once) -- balanced however uneven the lists.
    uint32_t key, seg = 0;
    if (BWD) {
        const uint32_t nunits = counters[2];
        if (blockIdx.x >= nunits) return;
        // units in DESCENDING key order: the row transform in front of this kernel wrote the planes in ascending order, the last
        // 256 MB of them are still in the memory-side cache (-1 %: the kernel is VALU-bound)
        const uint32_t unit = nunits - 1u - blockIdx.x;
        key = seg_tile[unit];
        seg = unit - seg_off[key];
    } else {
#ifdef FGS_SPLAT_ORDER_GROUPS
        key = tile_order ? tile_order[NP == 1 ? fgs_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x] : blockIdx.x;  // (b*P + p)*T + t
#else
        key = tile_order ? tile_order[blockIdx.x] : blockIdx.x;  // (b*P + p)*T + t
#endif
    }
    const uint32_t bp = key / tiles, t = key - bp * tiles;
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    const uint32_t X0 = tx * FGS_TILE, Y0 = ty * FGS_TILE;
    const uint32_t lane = threadIdx.x & 63u, lx = lane & 7u, ly = lane >> 3;
    const uint32_t wave = NP > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0u;
    const uint32_t wofs = wave * ACH;  // this wave's slice of the staging arrays
    float fx0 = (float)(X0 + lx), fy0 = (float)(Y0 + ly);
    asm("" : "+v"(fx0), "+v"(fy0));  // hoisted for good
    uint32_t start = ranges[2 * key] + seg * seg_len;
    uint32_t end = BWD ? min(ranges[2 * key + 1], start + seg_len) : ranges[2 * key + 1];
    if (NP > 1) {  // this wave's part of the list (whole chunks)
        const uint32_t per = ((end - start + NP * ACH - 1) / (NP * ACH)) * ACH;
        start = min(end, start + wave * per);
        end = min(end, start + per);
    }
    const size_t HW = (size_t)W * H;
    float2 *fbase = field + (size_t)bp * 3 * HW;  // [b][p][c][y][x]
    float re[4][3], im[4][3];  // FWD: accumulators.  BWD: field gradient at this lane's pixels
    float wd[4], ww[4];        // WAVE: sum a*depth, sum a   (BWD: their gradients)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint32_t px = X0 + 8u * (s & 1) + lx, py = Y0 + 8u * (s >> 1) + ly;
        wd[s] = 0.0f; ww[s] = 0.0f;
        if (WAVE && BWD && px < W && py < H) {
            const float2 g = dw[(size_t)bp * HW + (size_t)py * W + px];
            wd[s] = g.x; ww[s] = g.y;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            re[s][c] = 0.0f; im[s][c] = 0.0f;
            if (BWD && px < W && py < H) {
                const float2 g = fbase[(size_t)c * HW + (size_t)py * W + px];
                re[s][c] = g.x; im[s][c] = g.y;
            }
        }
    }
    for (uint32_t base = start; base < end; base += ACH) {
        const uint32_t n = min((uint32_t)ACH, end - base);
        if (lane < n) {
            const uint32_t gid = dup_ids[base + lane];
            const float4 *r = reinterpret_cast<const float4 *>(rec + (size_t)gid * FGS_REC_FLOATS);
            const float4 q0 = r[0], q1 = r[1], q2 = r[2];
            const uint32_t bbx = __float_as_uint(q2.z), bby = __float_as_uint(q2.w);
            // touched sub-tiles + the pixel bits of the tile (bit i: column X0 + i inside the bbox, bit 16 + i: row Y0 + i): in the
            // list loop a lane turns its column / row bits into all-ones / zero masks (v_bfe_i32) and and-s them onto G -- no
            // per-pixel compare / select, as on the blend path (issue costs: DESIGN.md section 4)
            uint32_t sflags, pbits;
            stage_decode(X0, Y0, bbx, bby, 1.0f, sflags, pbits);
            shm[wofs + lane] = sflags & 15u;
            if (BWD) {
                const uint32_t tx0 = (bbx & 0xFFFFu) / FGS_TILE, tx1 = ((bbx >> 16) - 1) / FGS_TILE, ty0 = (bby & 0xFFFFu) / FGS_TILE;
                she[lane] = dup_off[gid] + (ty - ty0) * (tx1 - tx0 + 1) + (tx - tx0);
            }
            const float4 *pz = reinterpret_cast<const float4 *>(ccs + (size_t)gid * 8);
            const float4 z0 = pz[0], z1 = pz[1];  // cc[0..2], cs[0] | cs[1..2]
            // conic pre-multiplied by K = -log2(e) / 2: G = exp2(K m) without a multiply per pixel (the backward's
            // dL/dconic = -1/2 dG/dm' ... is formed from the unscaled moments, below)
            sh0[wofs + lane] = make_float4(q0.x, q0.y, q0.z * NEG_HALF_LOG2E, q0.w * NEG_HALF_LOG2E);  // u, v, K ca, K cbc
            sh1[wofs + lane] = make_float4(q1.x * NEG_HALF_LOG2E, q1.y, __uint_as_float(pbits), 0.0f);   // K cd, op, pixel bits
            sh2[wofs + lane] = z0;
            sh3[wofs + lane] = make_float4(z1.x, z1.y, q2.y, 0.0f);  // .z = depth (WAVE)
        }
        __builtin_amdgcn_wave_barrier();  // wave-private staging: one wave's LDS instructions execute in order
        for (uint32_t j = 0; j < n; ++j) {
            const float4 q0 = sh0[wofs + j], q1 = sh1[wofs + j], q2 = sh2[wofs + j], q3 = sh3[wofs + j];
            const uint32_t msk = __builtin_amdgcn_readfirstlane(shm[wofs + j]);
            const uint32_t pbits = __float_as_uint(q1.z);
            const uint32_t mxs[2] = {(uint32_t)__builtin_amdgcn_sbfe((int)pbits, lx, 1), (uint32_t)__builtin_amdgcn_sbfe((int)pbits, lx + 8u, 1)};
            const uint32_t mys[2] = {(uint32_t)__builtin_amdgcn_sbfe((int)pbits, 16u + ly, 1), (uint32_t)__builtin_amdgcn_sbfe((int)pbits, 24u + ly, 1)};
            const float ca = q0.z, cbc = q0.w, cd = q1.x, op = q1.y;
            const float cc[3] = {q2.x, q2.y, q2.z}, cs[3] = {q2.w, q3.x, q3.y};
            float v_u = 0, v_v = 0, v_ca = 0, v_cbc = 0, v_cd = 0, v_op = 0, v_dep = 0;
            float v_cc[3] = {0, 0, 0}, v_cs[3] = {0, 0, 0};
            const float dz = q3.z;
            // the lane's two column / row offsets once per entry (no int -> float conversion in the list loop)
            const float dxs[2] = {fx0 - q0.x, fx0 + 8.0f - q0.x}, dys[2] = {fy0 - q0.y, fy0 + 8.0f - q0.y};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (!((msk >> s) & 1u)) continue;
                const float dx = dxs[s & 1], dy = dys[s >> 1];
                // (this unit is compiled with -ffp-contract=off -- build.py -- so the FMAs of the two VALU-bound loops, this one and
                // the column butterflies of fgs_colfft.h, are written out: sums of same-signed or well-separated terms, where
                // fusing is harmless; measured: dL/dlambda of K5 / G9 / G16 unchanged, config 5 back from 1.99 to 1.89 ms)
                const float m = fmaf(ca * dx, dx, fmaf(cbc * dx, dy, (cd * dy) * dy));  // K m
                const float G = __uint_as_float(__float_as_uint(__builtin_amdgcn_exp2f(m)) & (mxs[s & 1] & mys[s >> 1]));
                const float a = G * op;  // amplitude, DR:1270-1271 (no clamp on this path)
                if (!BWD) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { re[s][c] = fmaf(a, cc[c], re[s][c]); im[s][c] = fmaf(a, cs[c], im[s][c]); }
                    if (WAVE) { wd[s] = fmaf(a, dz, wd[s]); ww[s] += a; }  // DR:890-891
                } else {
                    float da = 0.0f;
                    if (WAVE) { da = fmaf(wd[s], dz, ww[s]); v_dep = fmaf(a, wd[s], v_dep); }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        da = fmaf(cc[c], re[s][c], fmaf(cs[c], im[s][c], da));
                        v_cc[c] = fmaf(a, re[s][c], v_cc[c]); v_cs[c] = fmaf(a, im[s][c], v_cs[c]);
                    }
                    // moments of t = dL/da G about the Gaussian's mean: {1, dx, dy, dx^2, dx dy, dy^2}.  The chain through
                    // a = G op and m (dL/dm = -1/2 t op; dL/d(u, v) = -dL/dm (2 ca dx + cbc dy, cbc dx + 2 cd dy), linear in the
                    // first moments) is applied once per Gaussian, in double, by k_project_bwd: 7 VALU per pass instead of 13
                    const float t = da * G;
                    v_op += t;
                    const float tx = t * dx, ty = t * dy;
                    v_u += tx; v_v += ty;
                    v_ca = fmaf(tx, dx, v_ca); v_cbc = fmaf(tx, dy, v_cbc); v_cd = fmaf(ty, dy, v_cd);
                }
            }
            if (BWD) {
                constexpr int NV = WAVE ? 13 : 12;
                float vals[NV] = {v_u, v_v, v_ca, v_cbc, v_cd, v_op, v_cc[0], v_cc[1], v_cc[2], v_cs[0], v_cs[1], v_cs[2]};
                if (WAVE) vals[NV - 1] = v_dep;
                const float tot = wave_sum_addtid<NV>(red, vals, lane);  // conflict-free parking, as in the blend backward
                const uint32_t e = she[j];
                if ((lane & 3u) == 3u && lane < 4u * NV && e < dcap)
                    grad_rows[(size_t)e * (WAVE ? 16 : FGS_GROW_FLOATS) + (lane >> 2)] = tot;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!BWD && NP > 1) {
        // add the partial fields in part order (wave 0 = part 0 accumulates parts 1, 2, ...): deterministic
        constexpr int NF = WAVE ? 32 : 24;
        if (wave != 0) {
            float *pp = part + ((size_t)(wave - 1) * NF) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { pp[(s * 6 + c) * 64] = re[s][c]; pp[(s * 6 + 3 + c) * 64] = im[s][c]; }
                if (WAVE) { pp[(24 + 2 * s) * 64] = wd[s]; pp[(25 + 2 * s) * 64] = ww[s]; }
            }
        }
        __syncthreads();
        if (wave != 0) return;
        for (int w = 1; w < NP; ++w) {
            const float *pp = part + ((size_t)(w - 1) * NF) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { re[s][c] += pp[(s * 6 + c) * 64]; im[s][c] += pp[(s * 6 + 3 + c) * 64]; }
                if (WAVE) { wd[s] += pp[(24 + 2 * s) * 64]; ww[s] += pp[(25 + 2 * s) * 64]; }
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t px = X0 + 8u * (s & 1) + lx, py = Y0 + 8u * (s >> 1) + ly;
            if (px < W && py < H) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    fbase[(size_t)c * HW + (size_t)py * W + px] = make_float2(re[s][c], im[s][c]);
                if (WAVE) dw[(size_t)bp * HW + (size_t)py * W + px] = make_float2(wd[s], ww[s]);
            }
        }
    }
}

// acc[b][c][k] = sum_p F[b][p][c][k] * H[c][p][k]
__global__ __launch_bounds__(256) void k_asm_accumulate(size_t HW, int B, int P, const float2 *__restrict__ field,
                                                        const float2 *__restrict__ htab,
                                                        float2 *__restrict__ acc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * 3 * HW) return;
    const size_t k = i % HW;
    const int c = (int)((i / HW) % 3), b = (int)(i / (3 * HW));
    float2 s = make_float2(0.0f, 0.0f);
    for (int p = 0; p < P; ++p) {
        const float2 f = field[(((size_t)b * P + p) * 3 + c) * HW + k];
        const float2 h = htab[((size_t)c * P + p) * HW + k];
        const float2 t = cmul(f, h);
        s.x += t.x; s.y += t.y;
    }
    acc[i] = s;
}

// ---- per-image scalars (maximum, dL/dM, number of maxima, dL/dlambda) as TWO-LEVEL reductions -------------------
// One partial per block, RED_BLOCKS blocks per image; the consumers' blocks fold the partials themselves (one value per
// thread, fixed order).  Round 1 reduced these with one device-scope atomic per wave on a single address per image;
// such atomics serialise at ~50 ns each on this part: k_asm_output_bwd1 took 54 us for one image and 419 us for
// eight, k_asm_max 26 / 191 us -- both read 6-9 MB per image -- and the float sums came out in arrival order.

__device__ __forceinline__ float block_max_256(float v) {  // all 256 threads call; every thread gets the result
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();  // (protects wmax against the previous use)
    if ((threadIdx.x & 63u) == 0) wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}
__device__ __forceinline__ float2 block_sum2_256(float a, float b) {  // fixed order: xor tree, then waves 0..3
    __shared__ float2 wsum[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = make_float2(a, b);
    __syncthreads();
    return make_float2((wsum[0].x + wsum[1].x) + (wsum[2].x + wsum[3].x), (wsum[0].y + wsum[1].y) + (wsum[2].y + wsum[3].y));
}
// maximum of an image's RED_BLOCKS block maxima (every thread of the block gets it)
__device__ __forceinline__ float image_max(const float *__restrict__ pmax, int b) {
    static_assert(RED_BLOCKS <= 256, "one partial per thread");
    return block_max_256(threadIdx.x < RED_BLOCKS ? pmax[(size_t)b * RED_BLOCKS + threadIdx.x] : 0.0f);
}

// ---- column-fused 2-D transforms (power-of-two heights) ------------------------------------------------------------
// rocFFT's 2-D C2C transform of the B*P*3 plane fields is a row kernel (0.30 ms for 805 MB at 8 images, 5.4 TB/s) and a
// column kernel (0.80 ms, 2 TB/s), followed here by k_asm_accumulate, which reads the spectra once more (0.23 ms).  For
// H = 64 ... 1024 (powers of two; widths that are whole column tiles) the column direction is done by our own kernels instead, fused
// with what follows / precedes it:
//   forward   rocFFT 1-D rows, then k_colfft_fwd: per (image, channel, tile of TC columns) and plane -- the H x TC tile streams into
//             LDS (rows of TC complex: 128-byte segments), radix-8 decimation-in-frequency FFT down the columns (first and, for
//             H = 8^k, last pass in registers; output in bit-reversed row order, undone by the store addresses), and the spectrum
//             goes into the Horner sums of the plane recurrence H_(p+1) = H_p D (no transfer-function value is read per plane);
//             acc_c and Z_c -- all the backward needs for dL/dlambda -- are written once;
//   backward  k_colfft_bwd: gF = gAcc conj(H_pc) per plane (the same recurrence) in bit-reversed row order, radix-8
//             decimation-in-time inverse FFT (natural order out) straight to HBM; the dL/dlambda terms from Z; then rocFFT
//             1-D inverse rows.
// The column pass reads the plane data once (forward) / writes it once (backward) instead of two reads and one write plus
// the accumulate kernel's pass.  Unnormalised, like hipFFT.  Twiddles w_N^n = exp(-2 pi i n / N) from a global table (k_asm_prep).
// Eight tile elements per thread: NT = N * TC / 8 threads per block (1024 for a 512 x 16 tile).  (The first version ran
// 256 threads with 32 elements each: 270 / 458 VGPRs, one wave per SIMD, 1.9 / 1.3 ms at 8 images.)  History and measurements:
// DESIGN_LOG.md 10.5.
constexpr int COLFFT_PER = 8;

template <int NT>
__device__ __forceinline__ void load_twiddles(float2 *tw, const float2 *__restrict__ tw_g, int n) {
    for (int i = threadIdx.x; i < n; i += NT) tw[i] = tw_g[i];
}


// Addressing of the column kernels: buffer loads / stores -- a UNIFORM base (image / plane / channel: a buffer resource in scalar
// registers, rebuilt per plane by scalar instructions), a uniform 32-bit byte offset per tile element (scalar) and a 32-bit per-thread
// byte offset that is computed once: no vector arithmetic per access.  (The first version recomputed 64-bit vector addresses per
// element and plane -- a fifth of the kernels' vector instructions; plain pointers with a 32-bit index are turned back into
// per-element 64-bit vector addresses by the compiler.)  Offsets stay below 2^31: a plane of one channel is H W 8 <= 2^23 W bytes.
typedef float colfft_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t colfft_rsrc(const void *ubase) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ubase), 0, 0x7fffffff, 0x00020000);  // raw buffer, no swizzle
}
__device__ __forceinline__ float2 ld_f2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const colfft_v2f v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st_f2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float2 v) {
    colfft_v2f t; t.x = v.x; t.y = v.y;
    __builtin_amdgcn_raw_buffer_store_b64(t, r, (int)voff, (int)soff, 0);
}
// frequency row held by the thread's e-th transformed element = FREQ_T (per thread) + colfft_freq_e (compile time), see k_colfft_fwd:
// INNER (rows 8 q + e): bitrev(8 q + e) = bitrev3(e) N/8 + bitrev(q); otherwise (rows q + e N/8): 8 bitrev(q) + bitrev3(e)
template <int LOGN, bool INNER>
__device__ __forceinline__ constexpr int colfft_freq_e(int e) {
    const int br3 = ((e & 1) << 2) | (e & 2) | ((e >> 2) & 1);
    return INNER ? br3 << (LOGN - 3) : br3;
}
template <int LOGN, bool INNER>
__device__ __forceinline__ int colfft_freq_t(int q) { return INNER ? bitrev<LOGN - 3>(q) : 8 * bitrev<LOGN - 3>(q); }

// The planes of image b that hold Gaussians, as a bit mask in scalar registers (bit p; every wave takes it with one ballot at kernel
// start): the column kernels used to ask seg_off plane by plane -- two dependent global loads in front of every plane step.
// (The interface admits at most 64 planes.)  No seg_off: every plane counts as occupied (skipping is an optimisation, empty planes hold zeros).
__device__ __forceinline__ uint64_t asm_plane_mask(const uint32_t *__restrict__ seg_off, int b, int P, uint32_t tiles) {
    if (!seg_off || P > 64) return ~0ull;
    const int l = (int)(threadIdx.x & 63u);
    const bool occ = l < P && !asm_plane_empty(seg_off, (uint32_t)(b * P + l), tiles);
    const uint64_t m = __ballot(occ);
    return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)m);
}
__device__ __forceinline__ int asm_plane_at_or_below(uint64_t mask, int pq, int p_lo) {  // last occupied plane in [p_lo, pq], or p_lo - 1
    if (pq < p_lo) return p_lo - 1;
    if (mask == ~0ull) return pq;  // every plane counts as occupied
    const uint64_t m = mask & (pq >= 63 ? ~0ull : ((2ull << pq) - 1ull)) & ~((1ull << p_lo) - 1ull);
    return m ? 63 - __builtin_clzll(m) : p_lo - 1;
}
__device__ __forceinline__ int asm_plane_at_or_above(uint64_t mask, int pq, int p_hi) {  // first occupied plane in [pq, p_hi), or p_hi
    if (pq >= p_hi) return p_hi;
    if (pq >= 64) return pq;
    const uint64_t m = mask & ~((1ull << pq) - 1ull);
    const int f = m ? __builtin_ctzll(m) : 64;
    return f < p_hi ? f : (p_hi > 64 ? 64 : p_hi);
}

// Launch order of the column kernels' blocks (grid (column tiles, 3, images x plane groups)) -> (column tile, channel, image, group).
// ORDER 0: as dispatched (round-robin over the XCDs).  1: every XCD walks a contiguous range, column tile fastest.  2: contiguous
// range, IMAGE fastest (from the time when every image re-read the transfer functions per plane: a table tile was then fetched into
// one XCD's L2 once and found there by the other images' blocks).  3 / 4: orders 1 / 0 with the images in DESCENDING order -- rocFFT's row
// pass walks the planes in ascending order, so the forward kernel starts on the planes written last and the backward kernel writes
// last what the row pass will read first: what is still in the 256 MB memory-side cache is not fetched from HBM.
// Measured (config 5, 8 images, profiles/r03_ab_config5_colfft.txt; k_colfft_fwd / k_colfft_bwd / the inverse row pass behind it, us):
// order 2: 224 / 152 / 286; order 3: 228 / 147 / 281; order 4: 221 / 144 / 256.  At one image (six plane groups, everything fits the cache)
// order 0 is 20 % faster than the XCD-contiguous orders: the kernels take ORDER_MANY from four images up.
template <int ORDER_MANY>
__device__ __forceinline__ void colfft_block(int PG, int &bx, int &c, int &b, int &grp) {
    const int ORDER = gridDim.z / (uint32_t)PG >= 4u ? ORDER_MANY : 0;
    if (ORDER == 4) {  // dispatch order, images descending
        bx = blockIdx.x; c = blockIdx.y; const int bz = (int)gridDim.z - 1 - (int)blockIdx.z; b = bz / PG; grp = bz - b * PG; return;
    }
    if (ORDER == 0) { bx = blockIdx.x; c = blockIdx.y; b = blockIdx.z / PG; grp = blockIdx.z - b * PG; return; }
    const uint32_t lin = fgs_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + 3u * blockIdx.z), gridDim.x * 3u * gridDim.z);
    if (ORDER == 1 || ORDER == 3) {  // 3: images in descending order (the planes rocFFT wrote last / reads first are handled first / last)
        bx = (int)(lin % gridDim.x); c = (int)((lin / gridDim.x) % 3u);
        int bz = (int)(lin / (gridDim.x * 3u));
        if (ORDER == 3) bz = (int)gridDim.z - 1 - bz;
        b = bz / PG; grp = bz - b * PG;
    } else {
        const uint32_t nimg = gridDim.z / (uint32_t)PG;
        b = (int)(lin % nimg); bx = (int)((lin / nimg) % gridDim.x); c = (int)((lin / (nimg * gridDim.x)) % 3u);
        grp = (int)(lin / (nimg * gridDim.x * 3u));
    }
}
#ifndef FGS_COLFFT_ORDER_FWD
#define FGS_COLFFT_ORDER_FWD 4
#endif
#ifndef FGS_COLFFT_ORDER_BWD
#define FGS_COLFFT_ORDER_BWD 4
#endif

// forward: acc[b][c] = sum_p F_pc H_pc and Z[b][c] = sum_p z_p F_pc H_pc.  grid (column tiles, 3, B)
// The spectra F are NOT stored (round 3): the backward needs them only in dL/dlambda_c = sum_k 2 pi dkz_k sum_p z_p dL/dtheta_pk with
// dL/dtheta_pk = -Im(conj(gAcc_k) H_pk F_pk), i.e. in Z_k = sum_p z_p H_pk F_pk -- linear in F, so it is summed here beside acc
// and the backward reads 50 MB of Z instead of 0.8 GB of spectra, which the forward no longer writes either.
// Planes of image b without any list entry are skipped (`seg_off`, see asm_plane_empty): their fields are zero (the row-fused build never writes them)
template <int LOGN, int TC>
__global__ __launch_bounds__((1 << LOGN) * TC / COLFFT_PER) void k_colfft_fwd(int W, int P, int PG, float2 *__restrict__ field,
                                                                           const float2 *__restrict__ htab,
                                                                           const float2 *__restrict__ tw_g,
                                                                           float2 *__restrict__ acc, float2 *__restrict__ zsum,
                                                                           float near_, float far_, float focal,
                                                                           const uint32_t *__restrict__ seg_off, uint32_t tiles) {
    // PG plane groups per image (launches of few images: more blocks, each summing its planes into its own partial
    // acc[(b, group)]; k_sum_groups adds them up): blockIdx.z = b * PG + group
    constexpr int N = 1 << LOGN, PER = COLFFT_PER, NT = N * TC / PER, E1 = N / 8;
    constexpr bool INNER = lds_fft_inner_in_registers<LOGN>();
    constexpr int NW = NT / 64, RPI = 128 / TC;  // waves; tile rows covered by one 64-lane x 16-byte load
    static_assert(PER == 8 && N / RPI == 4 * NW, "one 8-point butterfly per thread and pass; four tile loads per wave");
    // TWO tile buffers: while plane p is transformed in one, plane p + 1 streams from HBM straight into the other
    // (buffer_load ... lds: no registers, a whole plane step of latency cover -- the register prefetch of the first version had one
    // tile of 64 KB in flight per CU for part of the step and the kernel sat at 2.7 TB/s with every arithmetic instruction removed)
    __shared__ __attribute__((aligned(16))) float2 xa[N][TC];
    __shared__ __attribute__((aligned(16))) float2 xb[N][TC];
    __shared__ float2 tw[N / 2];
    load_twiddles<NT>(tw, tw_g, N / 2);
    int bx, c, b, grp;
    colfft_block<FGS_COLFFT_ORDER_FWD>(PG, bx, c, b, grp);
    const int bz = b * PG + grp, c0 = bx * TC;
    const int ppg = (P + PG - 1) / PG, p_lo = grp * ppg, p_hi = min(P, p_lo + ppg);
    const size_t HW = (size_t)N * W;
    // The thread's tile elements: rows q + e N/8 of its column in the opening block-size-N butterfly (LDS -> registers -> LDS), and
    // (LDS row r holds frequency bitrev(r) at the end) for N = 8^k the eight consecutive rows 8 q + e of the closing block-size-8
    // butterfly, in registers (its twiddles are all 1) straight into the epilogue; otherwise the rows q + e N/8 once more.
    const int col = threadIdx.x % TC, q = threadIdx.x / TC;  // (W is a multiple of TC: make_asm_plan)
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto out_row = [&](int e) { return INNER ? 8 * q + e : q + e * E1; };
    // NO transfer-function value is read per plane: the planes are equally spaced, H_p = H_lo D^(p - lo) (lo = the group's first
    // plane), so   acc = sum_p H_p F_p = H_lo S   and   Z = sum_p z_p H_p F_p = H_lo (z_lo S - step T)   with the Horner sums
    //     S_k = F_k + D S_(k+1),   T_k = D (T_(k+1) + S_(k+1))      [S_k = sum_(p>=k) D^(p-k) F_p, T_k = sum_(p>=k) (p-k) D^(p-k) F_p]
    // over the planes in DESCENDING order: D once per block, H_lo once at the end -- the table used to be re-read by every image
    // (as many bytes per step as the plane data, 45 us of this kernel and 107 us of the backward at 8 images even out of L2).
    float2 S[PER], T[PER], D[PER];
    const uint32_t off_out = (uint32_t)(colfft_freq_t<LOGN, INNER>(q) * W + col) * 8u;  // element e's frequency: + colfft_freq_e(e) W
    {
        const __amdgpu_buffer_rsrc_t dt = colfft_rsrc(htab + ((size_t)3 * P + c) * HW + c0);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            S[e] = T[e] = make_float2(0.0f, 0.0f);
            D[e] = ld_f2(dt, off_out, (uint32_t)(colfft_freq_e<LOGN, INNER>(e) * W) * 8u);
        }
    }
    const uint64_t occupied = asm_plane_mask(seg_off, b, P, tiles);
    auto prev_plane = [&](int pq) { return asm_plane_at_or_below(occupied, pq, p_lo); };  // last plane <= pq of this group with Gaussians
    auto skip_planes = [&](int m) {  // m planes with F = 0
        for (int i = 0; i < m; ++i) {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                T[e] = cmul(D[e], make_float2(T[e].x + S[e].x, T[e].y + S[e].y));
                S[e] = cmul(D[e], S[e]);
            }
        }
    };
    int p = prev_plane(p_hi - 1);
    const uint32_t off_tile = (uint32_t)((lane / (TC / 2)) * W + 2u * (lane % (TC / 2))) * 8u;  // lane's 16 bytes within RPI rows of the tile
    const float2 *f0 = field + ((size_t)b * P * 3 + c) * HW + c0;                                 // + plane * 3 HW
    typedef __attribute__((address_space(3))) void lds_void;
    auto request_tile = [&](float2 (*x)[TC], int plane) {  // lane i's 16 bytes land at (LDS base) + 16 i: RPI whole tile rows per load
        const __amdgpu_buffer_rsrc_t f = colfft_rsrc(f0 + (size_t)plane * 3 * HW);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t r0 = (wave + (uint32_t)NW * k) * RPI;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(f, (lds_void *)(uintptr_t)&x[r0][0], 16, (int)off_tile, (int)(r0 * W * 8u), 0, 0);
        }
    };
    // One plane in buffer x (PARITY picks it at compile time; the loop below is unrolled by two and peeled).  Barriers order LDS only
    // (lds_only_barrier): the next tile's four loads stay in flight through the whole step.
    auto plane_step = [&](auto has_next, auto parity, int pn) {
        float2 (*x)[TC] = decltype(parity)::value ? xb : xa;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the tile has landed ...
        lds_only_barrier();                               // ... everybody's has, and the other buffer has been read out
        if (decltype(has_next)::value) request_tile(decltype(parity)::value ? xa : xb, pn);
        float2 F[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) F[e] = x[q + e * E1][col];
        oct_dif<false>(F, tw[q], tw[2 * q], tw[4 * q]);  // block size N, points q + e N/8: w_N^q, w_N^2q, w_N^4q
#pragma unroll
        for (int e = 0; e < PER; ++e) x[q + e * E1][col] = F[e];
        lds_only_barrier();
        lds_fft_columns<LOGN, TC, NT, false, true, INNER, true>(x, tw);
#pragma unroll
        for (int e = 0; e < PER; ++e) F[e] = x[out_row(e)][col];
        if (INNER) oct_dif<true>(F, F[0], F[0], F[0]);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            T[e] = cmul(D[e], make_float2(T[e].x + S[e].x, T[e].y + S[e].y));
            const float2 ds = cmul(D[e], S[e]);
            S[e] = make_float2(F[e].x + ds.x, F[e].y + ds.y);
        }
    };
    if (p >= p_lo) {
        request_tile(xa, p);
        for (int pn = prev_plane(p - 1);; ) {
            if (pn < p_lo) { plane_step(std::false_type{}, std::false_type{}, 0); break; }
            plane_step(std::true_type{}, std::false_type{}, pn);
            skip_planes(p - pn - 1);
            p = pn; pn = prev_plane(p - 1);
            if (pn < p_lo) { plane_step(std::false_type{}, std::true_type{}, 0); break; }
            plane_step(std::true_type{}, std::true_type{}, pn);
            skip_planes(p - pn - 1);
            p = pn; pn = prev_plane(p - 1);
        }
        skip_planes(p - p_lo);
    }
    // (a plane group beyond the last plane -- P not a multiple of the group size -- writes zeros: its H_lo is not in the table)
    const bool any = p_lo < p_hi;
    const __amdgpu_buffer_rsrc_t h = colfft_rsrc(htab + ((size_t)c * P + (any ? p_lo : 0)) * HW + c0);
    const __amdgpu_buffer_rsrc_t a = colfft_rsrc(acc + ((size_t)bz * 3 + c) * HW + c0);
    const __amdgpu_buffer_rsrc_t zo = colfft_rsrc(zsum + ((size_t)bz * 3 + c) * HW + c0);
    const float z_lo = focal - plane_depth(p_lo, P, near_, far_), step = P > 1 ? (far_ - near_) / (float)(P - 1) : 0.0f;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const uint32_t so = (uint32_t)(colfft_freq_e<LOGN, INNER>(e) * W) * 8u;
        const float2 hl = any ? ld_f2(h, off_out, so) : make_float2(0.0f, 0.0f);
        st_f2(a, off_out, so, cmul(hl, S[e]));
        st_f2(zo, off_out, so, cmul(hl, make_float2(z_lo * S[e].x - step * T[e].x, z_lo * S[e].y - step * T[e].y)));
    }
}

// total[b][c][k] = sum over the PG plane groups of part[b][group][c][k] (fixed order); blockIdx.z = 0: acc, 1: Z
__global__ __launch_bounds__(256) void k_sum_groups(size_t n_per_image, int PG, const float2 *__restrict__ part0,
                                                    float2 *__restrict__ total0, const float2 *__restrict__ part1,
                                                    float2 *__restrict__ total1) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n_per_image) return;
    const float2 *part = blockIdx.z ? part1 : part0;
    float2 *total = blockIdx.z ? total1 : total0;
    float2 s = part[((size_t)b * PG) * n_per_image + i];
    for (int g = 1; g < PG; ++g) {
        const float2 v = part[((size_t)b * PG + g) * n_per_image + i];
        s.x += v.x; s.y += v.y;
    }
    total[(size_t)b * n_per_image + i] = s;
}

// backward: gF_pc = gAcc_c conj(H_pc), inverse-transformed down the columns, written where the forward's fields were; the block's part
// of dL/dlambda_c -> pwl[c][block] from Z (k_colfft_fwd; k_asm_accumulate_bwd has the maths in its per-plane form).  grid (column tiles, 3, B)
template <int LOGN, int TC>
__global__ __launch_bounds__((1 << LOGN) * TC / COLFFT_PER) void k_colfft_bwd(
    int W, int P, int PG, float near_, float far_, float focal, float inv_ndx, float inv_ndy,
    const float *__restrict__ wavelengths, const float2 *__restrict__ gacc, const float2 *__restrict__ htab,
    const float2 *__restrict__ tw_g, float2 *__restrict__ field, double *__restrict__ pwl,
    const uint32_t *__restrict__ seg_off, uint32_t tiles, const float2 *__restrict__ zsum) {
    constexpr int N = 1 << LOGN, PER = COLFFT_PER, NT = N * TC / PER, E1 = N / 8;
    constexpr bool INNER = lds_fft_inner_in_registers<LOGN>();
    __shared__ float2 x[N][TC];
    __shared__ float2 tw[N / 2];
    __shared__ double wpart[NT / 64];
    load_twiddles<NT>(tw, tw_g, N / 2);
    int bx, c, b, grp;
    colfft_block<FGS_COLFFT_ORDER_BWD>(PG, bx, c, b, grp);
    const int bz = b * PG + grp, c0 = bx * TC;
    const int ppg = (P + PG - 1) / PG, p_lo = grp * ppg, p_hi = min(P, p_lo + ppg);
    const size_t HW = (size_t)N * W;
    // the mirror image of k_colfft_fwd's element assignment: IN at LDS rows in_row(e) (frequency bitrev(row)) -- for N = 8^k the eight
    // rows of the opening block-size-8 butterfly, in registers; OUT the natural rows q + e N/8 of the closing block-size-N butterfly,
    // in registers straight to HBM
    const int col = threadIdx.x % TC, q = threadIdx.x / TC;  // (W is a multiple of TC: make_asm_plan)
    auto in_row = [&](int e) { return INNER ? 8 * q + e : q + e * E1; };
    float2 g[PER];
    // dL/dlambda_c = sum_k 2 pi (sum_p z_p dL/dtheta_pk) d kz_k / d lambda, with sum_p z_p dL/dtheta_pk = -Im(conj(gAcc_k) Z_k): once per
    // (image, channel), so plane group 0 carries it -- before the plane loop, so that one float stays live across it
    const __amdgpu_buffer_rsrc_t ga = colfft_rsrc(gacc + ((size_t)b * 3 + c) * HW + c0);
    const __amdgpu_buffer_rsrc_t zg = colfft_rsrc(zsum + ((size_t)b * 3 + c) * HW + c0);
    const uint32_t off_in = (uint32_t)(colfft_freq_t<LOGN, INNER>(q) * W + col) * 8u;  // element e's frequency row: + colfft_freq_e(e) W
    const uint32_t off_out = (uint32_t)(q * W + col) * 8u;                             // transformed element e:    + e (N/8) W
    const float wl = wavelengths[c], il = 1.0f / wl;
    const float fx = fftfreq(c0 + col, W, inv_ndx);
    float gl = 0.0f;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int ky = colfft_freq_t<LOGN, INNER>(q) + colfft_freq_e<LOGN, INNER>(e);  // = bitrev(in_row(e))
        g[e] = ld_f2(ga, off_in, (uint32_t)(colfft_freq_e<LOGN, INNER>(e) * W) * 8u);
        if (grp == 0) {
            const float2 Z = ld_f2(zg, off_in, (uint32_t)(colfft_freq_e<LOGN, INNER>(e) * W) * 8u);
            const float fy = fftfreq(ky, N, inv_ndy);
            const float kz2 = fgs_kz2(il, fx, fy);
            const float dkz = kz2 > 0.0f ? -(il * il * il) / sqrtf(kz2) : 0.0f;
            gl += 6.28318530717958647692f * (g[e].y * Z.x - g[e].x * Z.y) * dkz;
        }
    }
    // a plane without Gaussians has a zero spectrum and nobody reads its gradient: skipped (block-uniform)
    const uint64_t occupied = asm_plane_mask(seg_off, b, P, tiles);
    auto next_plane = [&](int pq) { return asm_plane_at_or_above(occupied, pq, p_hi); };
    int p = next_plane(p_lo);
    // gF_p = gAcc conj(H_p) by recurrence, as the forward: w = gAcc conj(H) at the GROUP's first plane (the only planes of H the table
    // holds on this path), then w <- w conj(D) per plane (H_(p+1) = H_p D); no transfer-function value is read inside the plane loop
    float2 w[PER], Dc[PER];
    {
        const __amdgpu_buffer_rsrc_t hp = colfft_rsrc(htab + ((size_t)c * P + (p_lo < p_hi ? p_lo : 0)) * HW + c0);
        const __amdgpu_buffer_rsrc_t dt = colfft_rsrc(htab + ((size_t)3 * P + c) * HW + c0);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const uint32_t so = (uint32_t)(colfft_freq_e<LOGN, INNER>(e) * W) * 8u;
            const float2 hh = ld_f2(hp, off_in, so), d = ld_f2(dt, off_in, so);
            w[e] = cmul(g[e], make_float2(hh.x, -hh.y));
            Dc[e] = make_float2(d.x, -d.y);
        }
        for (int i = p_lo; i < p && i < p_hi; ++i) {  // empty planes in front of the first occupied one
#pragma unroll
            for (int e = 0; e < PER; ++e) w[e] = cmul(w[e], Dc[e]);
        }
    }
    __syncthreads();  // twiddles
    auto plane_step = [&](int gap) {  // gap: planes to the next one with Gaussians
        const __amdgpu_buffer_rsrc_t f = colfft_rsrc(field + (((size_t)b * P + p) * 3 + c) * HW + c0);
        float2 v[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) v[e] = w[e];
        for (int i = 0; i < gap; ++i) {
#pragma unroll
            for (int e = 0; e < PER; ++e) w[e] = cmul(w[e], Dc[e]);
        }
        if (INNER) oct_dit<true>(v, v[0], v[0], v[0]);
#pragma unroll
        for (int e = 0; e < PER; ++e) x[in_row(e)][col] = v[e];
        __syncthreads();
        lds_fft_columns<LOGN, TC, NT, true, true, INNER>(x, tw);
#pragma unroll
        for (int e = 0; e < PER; ++e) v[e] = x[q + e * E1][col];
        __syncthreads();  // the tile is read: the next plane may be written over it
        oct_dit<false>(v, tw[q], tw[2 * q], tw[4 * q]);
#pragma unroll
        for (int e = 0; e < PER; ++e) st_f2(f, off_out, (uint32_t)(e * E1 * W) * 8u, v[e]);
    };
    while (p < p_hi) {
        const int pn = next_plane(p + 1);
        plane_step(pn - p);
        p = pn;
    }
    // the block's 8 192 terms (eight per thread, added in fp32 above) are summed in DOUBLE from here on: dL/dlambda cancels to a
    // fraction of its terms, and an fp32 tree over 1 024 partials alone put it 1.3e-4 from the fp64 oracle on a 512-row scene
    double gld = (double)gl;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gld += __shfl_xor(gld, o, 64);
    if ((threadIdx.x & 63u) == 0) wpart[threadIdx.x >> 6] = gld;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < NT / 64; ++w) t += wpart[w];
        pwl[(size_t)c * (gridDim.x * gridDim.z) + bz * gridDim.x + bx] = t;
    }
}

// per-image max of sqrt(|U|^2 + 1e-8) over pixels and channels (DR:1316-1322): block maxima -> pmax[b][block]
__global__ __launch_bounds__(256) void k_asm_max(size_t HW, float inv_hw, const float2 *__restrict__ total,
                                                 float *__restrict__ pmax) {
    const int b = blockIdx.y;
    float mx = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 3 * HW; i += (size_t)gridDim.x * 256) {
        mx = fmaxf(mx, fgs_asm_amplitude(fgs_asm_intensity(total[(size_t)b * 3 * HW + i], inv_hw)));
    }
    mx = block_max_256(mx);
    if (threadIdx.x == 0) pmax[(size_t)b * RED_BLOCKS + blockIdx.x] = mx;
}

struct PixOut {
    float r[3], a[3], n[3], v[3], asum, ta, M;
};

__device__ __forceinline__ void pixel_forward(const float2 u[3], float inv_hw, float maxval, const float bg[3],
                                              PixOut &o) {
    o.M = maxval < 1.0f ? 1.0f : maxval;  // clamp(min=1), DR:1322
    o.asum = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float I = fgs_asm_intensity(u[c], inv_hw);
        o.r[c] = fgs_asm_amplitude(I);  // (compared for equality with the image's maximum in the backward: fgs_internal.h)
        o.a[c] = sqrtf(I);  // |U| of the complex field, DR:1327
        o.asum += o.a[c];
        const float q = o.r[c] / o.M;
        o.n[c] = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
    }
    o.ta = o.asum < 0.0f ? 0.0f : (o.asum > 1.0f ? 1.0f : o.asum);
#pragma unroll
    for (int c = 0; c < 3; ++c) o.v[c] = o.n[c] + bg[c] * (1.0f - o.ta);
}

__global__ __launch_bounds__(256) void k_asm_output(size_t HW, float inv_hw, float bg0, float bg1, float bg2,
                                                    const float2 *__restrict__ total,
                                                    const float *__restrict__ pmax, float *__restrict__ scal,
                                                    float *__restrict__ out, const uint32_t *__restrict__ seg_off,
                                                    uint32_t lists_per_image) {
    const int b = blockIdx.y;
    const float maxval = image_max(pmax, b);
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[b] = maxval;  // kept for the backward
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float bg[3] = {bg0, bg1, bg2};
    if (asm_plane_empty(seg_off, (uint32_t)b, lists_per_image)) {
        // no visible Gaussian in this image (visible <=> at least one list entry): the reference returns the plain
        // background (DR:1207-1212), not sqrt(0 + 1e-8) pushed through the normalisation; all gradients are zero by themselves
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((size_t)b * 3 + c) * HW + i] = bg[c];
        return;
    }
    const float2 u[3] = {total[((size_t)b * 3 + 0) * HW + i], total[((size_t)b * 3 + 1) * HW + i],
                         total[((size_t)b * 3 + 2) * HW + i]};
    PixOut o;
    pixel_forward(u, inv_hw, maxval, bg, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = o.v[c];
        out[((size_t)b * 3 + c) * HW + i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
}

// backward of k_asm_output, pass 1: gM = dL/dM and the number of maxima (torch.max() spreads the gradient evenly over
// equal maxima): RED_BLOCKS block partials per image -> psum[b][block] = (gM, count)
__global__ __launch_bounds__(256) void k_asm_output_bwd1(size_t HW, float inv_hw, float bg0, float bg1, float bg2,
                                                         const float2 *__restrict__ total,
                                                         const float *__restrict__ scal, const float *__restrict__ g_out,
                                                         float2 *__restrict__ psum) {
    const int b = blockIdx.y;
    float gM = 0.0f, cnt = 0.0f;
    const float bg[3] = {bg0, bg1, bg2};
    const float maxval = scal[b];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
        const float2 u[3] = {total[((size_t)b * 3 + 0) * HW + i], total[((size_t)b * 3 + 1) * HW + i],
                             total[((size_t)b * 3 + 2) * HW + i]};
        PixOut o;
        pixel_forward(u, inv_hw, maxval, bg, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = o.v[c];
            const float gv = (v >= 0.0f && v <= 1.0f) ? g_out[((size_t)b * 3 + c) * HW + i] : 0.0f;
            const float q = o.r[c] / o.M;
            const float gq = (q >= 0.0f && q <= 1.0f) ? gv : 0.0f;
            gM -= gq * o.r[c] / (o.M * o.M);
            if (o.r[c] == maxval) cnt += 1.0f;
        }
    }
    const float2 t = block_sum2_256(gM, cnt);
    if (threadIdx.x == 0) psum[(size_t)b * RED_BLOCKS + blockIdx.x] = t;
}

// (gM, count) of image b from the block partials (every thread of the block gets them)
__device__ __forceinline__ float2 image_sums(const float2 *__restrict__ psum, int b) {
    const float2 v = threadIdx.x < RED_BLOCKS ? psum[(size_t)b * RED_BLOCKS + threadIdx.x] : make_float2(0.0f, 0.0f);
    return block_sum2_256(v.x, v.y);
}

// pass 2: gradient w.r.t. the (unnormalised) total field, written over `gtot`
__global__ __launch_bounds__(256) void k_asm_output_bwd2(size_t HW, float inv_hw, float bg0, float bg1, float bg2,
                                                         const float2 *__restrict__ total,
                                                         const float *__restrict__ scal,
                                                         const float2 *__restrict__ psum,
                                                         const float *__restrict__ g_out, float2 *__restrict__ gtot) {
    const int b = blockIdx.y;
    const float2 sums = image_sums(psum, b);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float2 u[3] = {total[((size_t)b * 3 + 0) * HW + i], total[((size_t)b * 3 + 1) * HW + i],
                         total[((size_t)b * 3 + 2) * HW + i]};
    const float bg[3] = {bg0, bg1, bg2};
    const float maxval = scal[b];
    const float cntm = sums.y;
    const float gMshare = (maxval >= 1.0f && cntm > 0.0f) ? sums.x / cntm : 0.0f;
    PixOut o;
    pixel_forward(u, inv_hw, maxval, bg, o);
    float gv[3], gta = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gv[c] = (o.v[c] >= 0.0f && o.v[c] <= 1.0f) ? g_out[((size_t)b * 3 + c) * HW + i] : 0.0f;
        gta -= gv[c] * bg[c];
    }
    const float gas = (o.asum >= 0.0f && o.asum <= 1.0f) ? gta : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float q = o.r[c] / o.M;
        float gr = ((q >= 0.0f && q <= 1.0f) ? gv[c] : 0.0f) / o.M;
        if (o.r[c] == maxval) gr += gMshare;
        // r = sqrt(I + 1e-8), a = sqrt(I), I = |u * inv_hw|^2
        const float gI = gr / (2.0f * o.r[c]) + (o.a[c] > 0.0f ? gas / (2.0f * o.a[c]) : 0.0f);
        const float k = 2.0f * gI * inv_hw * inv_hw;
        gtot[((size_t)b * 3 + c) * HW + i] = make_float2(k * u[c].x, k * u[c].y);
    }
}

// gF[b][p][c][k] = gAcc[b][c][k] * conj(H[c][p][k]) (overwrites the saved spectra F after using
// them for the wavelength gradient): theta = 2 pi z kz, dL/dtheta = -Im(H conj(gH)), gH = conj(F) gAcc
__global__ __launch_bounds__(256) void k_asm_accumulate_bwd(int W, int H, int B, int P, float near_, float far_,
                                                            float focal, float inv_ndx, float inv_ndy,
                                                            const float *__restrict__ wavelengths,
                                                            const float2 *__restrict__ gacc,
                                                            const float2 *__restrict__ htab,
                                                            float2 *__restrict__ field,
                                                            double *__restrict__ pwl /* [3][gridDim.x] block partials */) {
    const size_t HW = (size_t)W * H;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    double gl = 0.0;  // (summed in double from the first term on: see k_colfft_bwd)
    if (i < (size_t)B * HW) {
        const size_t k = i % HW;
        const int b = (int)(i / HW);
        const int kx = (int)(k % W), ky = (int)(k / W);
        const float fx = fftfreq(kx, W, inv_ndx), fy = fftfreq(ky, H, inv_ndy);
        const float wl = wavelengths[c];
        const float il = 1.0f / wl;
        const float kz2 = fgs_kz2(il, fx, fy);
        const float kz = kz2 > 0.0f ? sqrtf(kz2) : 0.0f;
        // d kz / d lambda = -1 / (lambda^3 kz) where kz2 > 0, else 0 (clamp / evanescent)
        const float dkz = kz2 > 0.0f ? -(il * il * il) / kz : 0.0f;
        const float2 g = gacc[((size_t)b * 3 + c) * HW + k];
        for (int p = 0; p < P; ++p) {
            const size_t fi = (((size_t)b * P + p) * 3 + c) * HW + k;
            const float2 f = field[fi];
            const float2 h = htab[((size_t)c * P + p) * HW + k];
            const float2 gH = cmul(make_float2(f.x, -f.y), g);          // conj(F) * gAcc
            const float gtheta = -(h.y * gH.x - h.x * gH.y);             // -Im(H * conj(gH))
            const float z = focal - plane_depth(p, P, near_, far_);
            gl += (double)(gtheta * (6.28318530717958647692f * z) * dkz);
            field[fi] = cmul(g, make_float2(h.x, -h.y));
        }
    }
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) gl += __shfl_xor(gl, of, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = gl;
    __syncthreads();
    if (threadIdx.x == 0) pwl[(size_t)c * gridDim.x + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// dL/dlambda_c = the sum of k_asm_accumulate_bwd's block partials, in a fixed order, in double (one block per channel)
__global__ __launch_bounds__(256) void k_asm_wavelength_grad(uint32_t n, const double *__restrict__ pwl,
                                                             float *__restrict__ g_wavelengths) {
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) s += pwl[(size_t)blockIdx.x * n + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) g_wavelengths[blockIdx.x] = (float)((part[0] + part[1]) + (part[2] + part[3]));
}


// ---------------------------------------------------------------------------------------------
// WaveFieldRenderer (DR:689-926): order-independent complex accumulation without
// depth planes or propagation; intensity -> sqrt -> per-image max normalisation -> background
// where the total amplitude is low; depth map = sum(a depth) / (sum a + 1e-8).
// ---------------------------------------------------------------------------------------------
struct WavePix {
    float r[3], n[3], v[3], tasq, ta, M;
};

__device__ __forceinline__ void wave_pixel_forward(const float2 u[3], float maxval, const float bg[3], WavePix &o) {
    o.M = maxval < 1.0f ? 1.0f : maxval;
    float isum = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float I = u[c].x * u[c].x + u[c].y * u[c].y;
        isum += I;
        o.r[c] = sqrtf(I + 1e-8f);                           // DR:898
        const float q = o.r[c] / o.M;                        // DR:902-905
        o.n[c] = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
    }
    o.tasq = sqrtf(isum + 1e-8f);                            // DR:908
    o.ta = o.tasq < 0.0f ? 0.0f : (o.tasq > 1.0f ? 1.0f : o.tasq);
#pragma unroll
    for (int c = 0; c < 3; ++c) o.v[c] = o.n[c] + bg[c] * (1.0f - o.ta);
}

__global__ __launch_bounds__(256) void k_wave_max(size_t HW, const float2 *__restrict__ field, float *__restrict__ pmax) {
    const int b = blockIdx.y;
    float mx = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 3 * HW; i += (size_t)gridDim.x * 256) {
        const float2 u = field[(size_t)b * 3 * HW + i];
        mx = fmaxf(mx, sqrtf(u.x * u.x + u.y * u.y + 1e-8f));
    }
    mx = block_max_256(mx);
    if (threadIdx.x == 0) pmax[(size_t)b * RED_BLOCKS + blockIdx.x] = mx;
}

__global__ __launch_bounds__(256) void k_wave_output(size_t HW, float bg0, float bg1, float bg2,
                                                     const float2 *__restrict__ field, const float2 *__restrict__ dw,
                                                     const float *__restrict__ pmax, float *__restrict__ scal,
                                                     float *__restrict__ out, float *__restrict__ out_depth,
                                                     const uint32_t *__restrict__ seg_off, uint32_t lists_per_image) {
    const int b = blockIdx.y;
    const float maxval = image_max(pmax, b);
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[b] = maxval;  // kept for the backward
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float bg[3] = {bg0, bg1, bg2};
    if (asm_plane_empty(seg_off, (uint32_t)b, lists_per_image)) {  // no visible Gaussian: plain background, zero depth (DR:801-808)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((size_t)b * 3 + c) * HW + i] = bg[c];
        out_depth[(size_t)b * HW + i] = 0.0f;
        return;
    }
    const float2 u[3] = {field[((size_t)b * 3 + 0) * HW + i], field[((size_t)b * 3 + 1) * HW + i],
                         field[((size_t)b * 3 + 2) * HW + i]};
    WavePix o;
    wave_pixel_forward(u, maxval, bg, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = o.v[c];
        out[((size_t)b * 3 + c) * HW + i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
    const float2 d = dw[(size_t)b * HW + i];
    out_depth[(size_t)b * HW + i] = d.x / (d.y + 1e-8f);      // DR:924
}

__global__ __launch_bounds__(256) void k_wave_output_bwd1(size_t HW, float bg0, float bg1, float bg2,
                                                          const float2 *__restrict__ field, const float *__restrict__ scal,
                                                          const float *__restrict__ g_out, float2 *__restrict__ psum) {
    const int b = blockIdx.y;
    float gM = 0.0f, cnt = 0.0f;
    const float bg[3] = {bg0, bg1, bg2};
    const float maxval = scal[b];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
        const float2 u[3] = {field[((size_t)b * 3 + 0) * HW + i], field[((size_t)b * 3 + 1) * HW + i],
                             field[((size_t)b * 3 + 2) * HW + i]};
        WavePix o;
        wave_pixel_forward(u, maxval, bg, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gv = (o.v[c] >= 0.0f && o.v[c] <= 1.0f) ? g_out[((size_t)b * 3 + c) * HW + i] : 0.0f;
            const float q = o.r[c] / o.M;
            const float gq = (q >= 0.0f && q <= 1.0f) ? gv : 0.0f;
            gM -= gq * o.r[c] / (o.M * o.M);
            if (o.r[c] == maxval) cnt += 1.0f;
        }
    }
    const float2 t = block_sum2_256(gM, cnt);
    if (threadIdx.x == 0) psum[(size_t)b * RED_BLOCKS + blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_wave_output_bwd2(size_t HW, float bg0, float bg1, float bg2,
                                                          const float2 *__restrict__ field,
                                                          const float2 *__restrict__ dw, const float *__restrict__ scal,
                                                          const float2 *__restrict__ psum,
                                                          const float *__restrict__ g_out,
                                                          const float *__restrict__ g_depth, float2 *__restrict__ gfield,
                                                          float2 *__restrict__ gdw) {
    const int b = blockIdx.y;
    const float2 sums = image_sums(psum, b);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float2 u[3] = {field[((size_t)b * 3 + 0) * HW + i], field[((size_t)b * 3 + 1) * HW + i],
                         field[((size_t)b * 3 + 2) * HW + i]};
    const float bg[3] = {bg0, bg1, bg2};
    const float maxval = scal[b];
    const float cntm = sums.y;
    const float gMshare = (maxval >= 1.0f && cntm > 0.0f) ? sums.x / cntm : 0.0f;
    WavePix o;
    wave_pixel_forward(u, maxval, bg, o);
    float gv[3], gta = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gv[c] = (o.v[c] >= 0.0f && o.v[c] <= 1.0f) ? g_out[((size_t)b * 3 + c) * HW + i] : 0.0f;
        gta -= gv[c] * bg[c];
    }
    // ta = clamp(sqrt(sum_c I_c + 1e-8), 0, 1): d ta / d I_c = 1 / (2 tasq) inside the clamp
    const float gIsum = (o.tasq >= 0.0f && o.tasq <= 1.0f) ? gta / (2.0f * o.tasq) : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float q = o.r[c] / o.M;
        float gr = ((q >= 0.0f && q <= 1.0f) ? gv[c] : 0.0f) / o.M;
        if (o.r[c] == maxval) gr += gMshare;
        const float gI = gr / (2.0f * o.r[c]) + gIsum;
        gfield[((size_t)b * 3 + c) * HW + i] = make_float2(2.0f * gI * u[c].x, 2.0f * gI * u[c].y);
    }
    // depth_map = Ad / (Wt + 1e-8)
    const float2 d = dw[(size_t)b * HW + i];
    const float gdm = g_depth[(size_t)b * HW + i];
    const float den = d.y + 1e-8f;
    gdw[(size_t)b * HW + i] = make_float2(gdm / den, -gdm * d.x / (den * den));
}

struct WavePlan {
    FgsWaveDims w;
    FgsPlan base;
    size_t HW, v_field, v_dw, v_scal, v_ccs, v_total_bytes, c_gfield, c_gdw, c_rows, c_part, c_total_bytes;
};

int make_wave_plan(const FgsWaveDims *w, WavePlan *p) {
    if (!w) { fgs_set_error("null dims"); return FGS_EINVAL; }
    if (w->phase_channels != 1 && w->phase_channels != 3) { fgs_set_error("invalid phase_channels"); return FGS_EINVAL; }
    FgsDims d{};
    d.batch = w->batch; d.num_gaussians = w->num_gaussians; d.width = w->width; d.height = w->height;
    d.max_radius = w->max_radius;
    for (int i = 0; i < 3; ++i) d.background[i] = w->background[i];
    d.num_cameras = w->num_cameras;
    p->w = *w;
    const int rc = fgs_make_plan(&d, &p->base, 1, false);
    if (rc) return rc;
    const size_t B = w->batch, HW = (size_t)w->width * w->height;
    p->HW = HW;
    size_t o = p->base.L.total_bytes;
    p->v_field = o; o = align256(o + B * 3 * HW * 8);
    p->v_dw = o; o = align256(o + B * HW * 8);
    p->v_scal = o; o = align256(o + B * 4 * 4);
    p->v_ccs = o; o = align256(o + B * (size_t)w->num_gaussians * 8 * 4);
    p->v_total_bytes = o;
    o = p->base.s_total;
    p->c_gfield = o; o = align256(o + B * 3 * HW * 8);
    p->c_gdw = o; o = align256(o + B * HW * 8);
    p->c_rows = o; o = align256(o + p->base.L.dup_capacity * 16 * 4);
    p->c_part = o; o = align256(o + B * RED_BLOCKS * 8);  // float2 [B][RED_BLOCKS] block partials of the per-image scalars
    p->c_total_bytes = o;
    return FGS_OK;
}

int check_ptrs(const void *const *ptrs, int n, const char *who) {
    for (int i = 0; i < n; ++i)
        if (!ptrs[i]) { fgs_set_error("%s: null pointer argument #%d", who, i); return FGS_EINVAL; }
    return FGS_OK;
}

}  // namespace

extern "C" {

int fgs_asm_workspace_bytes(const FgsAsmDims *dims, size_t *saved_bytes, size_t *scratch_bytes) {
    AsmPlan p;
    const int rc = make_asm_plan(dims, &p, true);
    if (rc) return rc;
    if (saved_bytes) *saved_bytes = p.v_total_bytes;
    if (scratch_bytes) *scratch_bytes = p.c_total_bytes;
    return FGS_OK;
}

int fgs_asm_forward(const FgsAsmDims *dims, const float *cameras, const float *pos, const float *scale,
                    const float *quat, const float *color, const float *opacity, const float *phase,
                    const float *wavelengths, float *out_rgb, void *saved, void *scratch, void *stream) {
    AsmPlan p;
    int rc = make_asm_plan(dims, &p, true);
    if (rc) return rc;
    const void *ptrs[] = {cameras, pos, scale, quat, color, opacity, phase, wavelengths, out_rgb, saved, scratch};
    if ((rc = check_ptrs(ptrs, 11, "fgs_asm_forward"))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *sv = reinterpret_cast<char *>(saved), *sc = reinterpret_cast<char *>(scratch);
    const FgsAsmDims &a = p.a;
    const int B = a.batch, P = a.num_planes, W = a.width, H = a.height;
    const size_t HW = p.HW;
    fgs_stage_begin(ST_PROJECT, st);
    if ((rc = fgs_launch_project(p.base, cameras, pos, scale, quat, color, opacity, sv, st, P, a.depth_near, a.depth_far)))
        return rc;
    fgs_stage_end(ST_PROJECT, st);
    if ((rc = fgs_launch_binning(p.base, sv, sc, st))) return rc;
    float2 *field = reinterpret_cast<float2 *>(sv + p.v_field);
    float2 *htab = reinterpret_cast<float2 *>(sv + p.v_htab);
    float2 *total = reinterpret_cast<float2 *>(sv + p.v_total);
    float *scal = reinterpret_cast<float *>(sv + p.v_scal);
    const uint32_t grid = (uint32_t)B * P * p.base.tiles;
    float *ccs = reinterpret_cast<float *>(sv + p.v_ccs);
    const uint32_t ngauss = (uint32_t)B * (uint32_t)a.num_gaussians;
    // torch.fft.fftfreq(n, d) = arange * (float)(1.0 / (n * d)) with d the reference's Python float (a double): DR:959-961
    const float inv_ndx = (float)(1.0 / ((double)W * a.pixel_pitch));
    const float inv_ndy = (float)(1.0 / ((double)H * a.pixel_pitch));
    {
        // the column kernels read H at the first plane of each plane group only (and D); the rocFFT 2-D path reads every plane
        const int pstride = p.col_logn ? (P + p.col_pg - 1) / p.col_pg : 1, np = (P + pstride - 1) / pstride;
        const size_t nh = 3 * (size_t)(np + 1) * (size_t)(W / 2 + 1) * (size_t)(H / 2 + 1);  // one quadrant, mirrored; + the step factors D
        const uint32_t nb_ph = (ngauss + 255) / 256, nb_tr = (uint32_t)((nh + 255) / 256);
        hipLaunchKernelGGL(k_asm_prep, dim3(nb_ph + nb_tr + 1), dim3(256), 0, st, nb_ph, nb_tr, ngauss, a.phase_channels,
                           color, phase, ccs, W, H, P, a.depth_near, a.depth_far, a.focal_depth, inv_ndx, inv_ndy,
                           wavelengths, htab, p.col_logn ? H : 0, reinterpret_cast<float2 *>(sv + p.v_tw), pstride);
        FGS_LAUNCH_CHECK("k_asm_prep");
    }
    // the column kernels skip the planes of an image that hold no Gaussian (their fields are zero), as the reference
    // skips them (DR:1302)
    const uint32_t *seg_off_d = reinterpret_cast<const uint32_t *>(sv + p.base.L.seg_off);
    fgs_stage_begin(ST_SPLAT_FWD, st);
#define FGS_SPLAT_FWD(WV, NPV, DW)                                                                                     \
    hipLaunchKernelGGL((k_asm_splat<false, WV, NPV>), dim3(grid), dim3(64 * NPV), 0, st, (uint32_t)p.base.tiles,       \
                       (uint32_t)p.base.L.tiles_x, (uint32_t)P, (uint32_t)W, (uint32_t)H,                              \
                       (uint32_t)p.base.L.dup_capacity, reinterpret_cast<const uint32_t *>(sv + p.base.L.tile_order),  \
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.ranges),                                       \
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.dup_ids),                                      \
                       reinterpret_cast<const float *>(sv + p.base.L.rec), ccs,                                        \
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.dup_off), field, (float *)nullptr, DW,         \
                       (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u)
    // one wave per list once the launch has enough lists to fill the chip, else four list parts per list
    if (grid >= ASM_ONE_WAVE_LISTS) FGS_SPLAT_FWD(false, 1, (float2 *)nullptr); else FGS_SPLAT_FWD(false, ASM_FWD_PARTS, (float2 *)nullptr);
    FGS_LAUNCH_CHECK("k_asm_splat");
    fgs_stage_end(ST_SPLAT_FWD, st);
    fgs_stage_begin(ST_FIELD_FWD, st);
    if (p.col_logn) {
        // rows by rocFFT, columns + transfer function + plane sum in one pass of our own (k_colfft_fwd)
        if ((rc = fgs_fft_rows_exec(W, B * P * 3 * H, field, HIPFFT_FORWARD, sc + p.c_fftwork, st))) return rc;
        float2 *tw = reinterpret_cast<float2 *>(sv + p.v_tw);
        float2 *accp = reinterpret_cast<float2 *>(sc + p.c_accp);
        const int PG = p.col_pg;
        float2 *zsum = reinterpret_cast<float2 *>(sv + p.v_zsum), *zsump = accp + (size_t)B * PG * 3 * HW;
#define FGS_COLFFT_FWD(LG, TCV)                                                                                       \
    hipLaunchKernelGGL((k_colfft_fwd<LG, TCV>), dim3((unsigned)((W + TCV - 1) / TCV), 3, B * PG),                    \
                       dim3((1 << LG) * TCV / COLFFT_PER), 0, st, W, P, PG, field, htab, tw, PG > 1 ? accp : total,         \
                       PG > 1 ? zsump : zsum, a.depth_near, a.depth_far, a.focal_depth, seg_off_d, (uint32_t)p.base.tiles)
        switch (p.col_logn) {
            case 6: FGS_COLFFT_FWD(6, 16); break;
            case 7: FGS_COLFFT_FWD(7, 16); break;
            case 8: FGS_COLFFT_FWD(8, 16); break;
            case 9: FGS_COLFFT_FWD(9, FGS_COLFFT_TC9); break;
            default: FGS_COLFFT_FWD(10, 8); break;
        }
#undef FGS_COLFFT_FWD
        FGS_LAUNCH_CHECK("k_colfft_fwd");
        if (PG > 1) {
            hipLaunchKernelGGL(k_sum_groups, dim3((unsigned)((3 * HW + 255) / 256), B, 2), dim3(256), 0, st, 3 * HW, PG, accp, total,
                               zsump, zsum);
            FGS_LAUNCH_CHECK("k_sum_groups");
        }
    } else {
        if ((rc = fgs_fft_exec(H, W, B * P * 3, field, HIPFFT_FORWARD, sc + p.c_fftwork, st))) return rc;
        const size_t na = (size_t)B * 3 * HW;
        hipLaunchKernelGGL(k_asm_accumulate, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, HW, B, P, field, htab,
                           total);
        FGS_LAUNCH_CHECK("k_asm_accumulate");
    }
    const float inv_hw = 1.0f / (float)HW;
    float *pmax = reinterpret_cast<float *>(sc + p.c_part);
    // the inverse transform's column pass leaves the per-image amplitude maxima as block partials (power-of-two heights whose
    // column tiles fit the RED_BLOCKS slots); otherwise k_asm_max reduces them in a launch of its own
    bool max_fused = false;
    if ((rc = fgs_fft2_inverse_with_max(H, W, B, total, sc + p.c_fftwork, pmax, RED_BLOCKS, inv_hw, &max_fused, st))) return rc;
    if (!max_fused) {
        hipLaunchKernelGGL(k_asm_max, dim3(RED_BLOCKS, B), dim3(256), 0, st, HW, inv_hw, total, pmax);
        FGS_LAUNCH_CHECK("k_asm_max");
    }
    hipLaunchKernelGGL(k_asm_output, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, st, HW, inv_hw,
                       a.background[0], a.background[1], a.background[2], total, pmax, scal, out_rgb, seg_off_d,
                       (uint32_t)P * (uint32_t)p.base.tiles);
    FGS_LAUNCH_CHECK("k_asm_output");
    fgs_stage_end(ST_FIELD_FWD, st);
    return FGS_OK;
}

int fgs_asm_backward(const FgsAsmDims *dims, const float *cameras, const float *pos, const float *scale,
                     const float *quat, const float *color, const float *opacity, const float *phase,
                     const float *wavelengths, void *saved, void *scratch, const float *g_rgb, float *g_pos,
                     float *g_scale, float *g_quat, float *g_color, float *g_opacity, float *g_phase,
                     float *g_wavelengths, void *stream) {
    AsmPlan p;
    int rc = make_asm_plan(dims, &p, true);
    if (rc) return rc;
    const void *ptrs[] = {cameras, pos, scale, quat, color, opacity, phase, wavelengths, saved, scratch, g_rgb,
                          g_pos, g_scale, g_quat, g_color, g_opacity, g_phase, g_wavelengths};
    if ((rc = check_ptrs(ptrs, 18, "fgs_asm_backward"))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *sv = reinterpret_cast<char *>(saved), *sc = reinterpret_cast<char *>(scratch);
    const FgsAsmDims &a = p.a;
    const int B = a.batch, P = a.num_planes, W = a.width, H = a.height;
    const size_t HW = p.HW;
    float2 *field = reinterpret_cast<float2 *>(sv + p.v_field);
    float2 *htab = reinterpret_cast<float2 *>(sv + p.v_htab);
    float2 *total = reinterpret_cast<float2 *>(sv + p.v_total);
    float *scal = reinterpret_cast<float *>(sv + p.v_scal);
    float2 *gtot = reinterpret_cast<float2 *>(sc + p.c_acc);
    const float inv_hw = 1.0f / (float)HW;
    fgs_stage_begin(ST_FIELD_BWD, st);
    const dim3 gpix((unsigned)((HW + 255) / 256), B);
    float2 *psum = reinterpret_cast<float2 *>(sc + p.c_part);
    double *pwl = reinterpret_cast<double *>(sc + p.c_part + (size_t)B * RED_BLOCKS * 8);
    hipLaunchKernelGGL(k_asm_output_bwd1, dim3(RED_BLOCKS, B), dim3(256), 0, st, HW, inv_hw, a.background[0],
                       a.background[1], a.background[2], total, scal, g_rgb, psum);
    FGS_LAUNCH_CHECK("k_asm_output_bwd1");
    hipLaunchKernelGGL(k_asm_output_bwd2, gpix, dim3(256), 0, st, HW, inv_hw, a.background[0], a.background[1],
                       a.background[2], total, scal, psum, g_rgb, gtot);
    FGS_LAUNCH_CHECK("k_asm_output_bwd2");
    // adjoint of the unnormalised inverse FFT is the unnormalised forward FFT
    if ((rc = fgs_fft2_exec(H, W, B * 3, gtot, HIPFFT_FORWARD, sc + p.c_fftwork, st))) return rc;
    // torch.fft.fftfreq(n, d) = arange * (float)(1.0 / (n * d)) with d the reference's Python float (a double): DR:959-961
    const float inv_ndx = (float)(1.0 / ((double)W * a.pixel_pitch));
    const float inv_ndy = (float)(1.0 / ((double)H * a.pixel_pitch));
    unsigned nwl = (unsigned)(((size_t)B * HW + 255) / 256);
    if (p.col_logn) {
        // gF = gAcc conj(H) and the inverse transform down the columns in one pass (k_colfft_bwd), then rocFFT rows
        const float2 *tw = reinterpret_cast<const float2 *>(sv + p.v_tw);
#define FGS_COLFFT_BWD(LG, TCV)                                                                                       \
    do {                                                                                                              \
        nwl = (unsigned)((W + TCV - 1) / TCV) * (unsigned)(B * p.col_pg);                                             \
        hipLaunchKernelGGL((k_colfft_bwd<LG, TCV>), dim3((unsigned)((W + TCV - 1) / TCV), 3, B * p.col_pg),          \
                           dim3((1 << LG) * TCV / COLFFT_PER), 0, st, W, P, p.col_pg, a.depth_near, a.depth_far,      \
                           a.focal_depth, inv_ndx, inv_ndy, wavelengths, gtot, htab, tw, field, pwl,                  \
                           reinterpret_cast<const uint32_t *>(sv + p.base.L.seg_off),                                 \
                           (uint32_t)p.base.tiles, reinterpret_cast<const float2 *>(sv + p.v_zsum));                 \
    } while (0)
        switch (p.col_logn) {
            case 6: FGS_COLFFT_BWD(6, 16); break;
            case 7: FGS_COLFFT_BWD(7, 16); break;
            case 8: FGS_COLFFT_BWD(8, 16); break;
            case 9: FGS_COLFFT_BWD(9, FGS_COLFFT_TC9); break;
            default: FGS_COLFFT_BWD(10, 8); break;
        }
#undef FGS_COLFFT_BWD
        FGS_LAUNCH_CHECK("k_colfft_bwd");
    } else {
        hipLaunchKernelGGL(k_asm_accumulate_bwd, dim3(nwl, 3), dim3(256), 0, st, W, H, B, P, a.depth_near, a.depth_far,
                           a.focal_depth, inv_ndx, inv_ndy, wavelengths, gtot, htab, field, pwl);
        FGS_LAUNCH_CHECK("k_asm_accumulate_bwd");
    }
    hipLaunchKernelGGL(k_asm_wavelength_grad, dim3(3), dim3(256), 0, st, nwl, pwl, g_wavelengths);
    FGS_LAUNCH_CHECK("k_asm_wavelength_grad");
    // adjoint of the forward FFT is the unnormalised inverse FFT
    if (p.col_logn) {
        if ((rc = fgs_fft_rows_exec(W, B * P * 3 * H, field, HIPFFT_BACKWARD, sc + p.c_fftwork, st))) return rc;
    } else if ((rc = fgs_fft_exec(H, W, B * P * 3, field, HIPFFT_BACKWARD, sc + p.c_fftwork, st))) {
        return rc;
    }
    fgs_stage_end(ST_FIELD_BWD, st);
    fgs_stage_begin(ST_SPLAT_BWD, st);
    const uint32_t grid = (uint32_t)p.base.L.seg_capacity;  // depth-segment units
    float *rows = reinterpret_cast<float *>(sc + p.base.s_grows);
    hipLaunchKernelGGL((k_asm_splat<true, false, 1>), dim3(grid), dim3(64), 0, st, (uint32_t)p.base.tiles,
                       (uint32_t)p.base.L.tiles_x, (uint32_t)P, (uint32_t)W, (uint32_t)H,
                       (uint32_t)p.base.L.dup_capacity,
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.tile_order),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.ranges),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.dup_ids),
                       reinterpret_cast<const float *>(sv + p.base.L.rec), reinterpret_cast<const float *>(sv + p.v_ccs),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.dup_off), field, rows, (float2 *)nullptr,
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.counters),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.seg_off),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.seg_tile), (uint32_t)p.base.L.seg_len);
    FGS_LAUNCH_CHECK("k_asm_splat_bwd");
    fgs_stage_end(ST_SPLAT_BWD, st);
    fgs_stage_begin(ST_PROJECT_BWD, st);
    if ((rc = fgs_launch_asm_project_bwd(p.base, cameras, pos, scale, quat, color, phase, a.phase_channels, sv, rows,
                                         g_pos, g_scale, g_quat, g_color, g_opacity, g_phase, st)))
        return rc;
    fgs_stage_end(ST_PROJECT_BWD, st);
    return FGS_OK;
}

int fgs_wave_workspace_bytes(const FgsWaveDims *dims, size_t *saved_bytes, size_t *scratch_bytes) {
    WavePlan p;
    const int rc = make_wave_plan(dims, &p);
    if (rc) return rc;
    if (saved_bytes) *saved_bytes = p.v_total_bytes;
    if (scratch_bytes) *scratch_bytes = p.c_total_bytes;
    return FGS_OK;
}

int fgs_wave_forward(const FgsWaveDims *dims, const float *cameras, const float *pos, const float *scale,
                     const float *quat, const float *color, const float *opacity, const float *phase,
                     float *out_rgb, float *out_depth, void *saved, void *scratch, void *stream) {
    WavePlan p;
    int rc = make_wave_plan(dims, &p);
    if (rc) return rc;
    const void *ptrs[] = {cameras, pos, scale, quat, color, opacity, phase, out_rgb, out_depth, saved, scratch};
    if ((rc = check_ptrs(ptrs, 11, "fgs_wave_forward"))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *sv = reinterpret_cast<char *>(saved), *sc = reinterpret_cast<char *>(scratch);
    const int B = p.w.batch, W = p.w.width, H = p.w.height;
    const size_t HW = p.HW;
    fgs_stage_begin(ST_PROJECT, st);
    if ((rc = fgs_launch_project(p.base, cameras, pos, scale, quat, color, opacity, sv, st))) return rc;
    fgs_stage_end(ST_PROJECT, st);
    if ((rc = fgs_launch_binning(p.base, sv, sc, st))) return rc;
    fgs_stage_begin(ST_SPLAT_FWD, st);
    float2 *field = reinterpret_cast<float2 *>(sv + p.v_field);
    float2 *dw = reinterpret_cast<float2 *>(sv + p.v_dw);
    float *scal = reinterpret_cast<float *>(sv + p.v_scal);
    const uint32_t grid = (uint32_t)B * p.base.tiles, P = 1;
    float *ccs = reinterpret_cast<float *>(sv + p.v_ccs);
    const uint32_t ngauss = (uint32_t)B * (uint32_t)p.w.num_gaussians;
    hipLaunchKernelGGL(k_asm_phasors, dim3((ngauss + 255) / 256), dim3(256), 0, st, ngauss, p.w.phase_channels, color, phase, ccs);
    FGS_LAUNCH_CHECK("k_asm_phasors");
    if (grid >= ASM_ONE_WAVE_LISTS) FGS_SPLAT_FWD(true, 1, dw); else FGS_SPLAT_FWD(true, ASM_FWD_PARTS, dw);
#undef FGS_SPLAT_FWD
    FGS_LAUNCH_CHECK("k_wave_splat");
    fgs_stage_end(ST_SPLAT_FWD, st);
    fgs_stage_begin(ST_FIELD_FWD, st);
    float *pmax = reinterpret_cast<float *>(sc + p.c_part);
    hipLaunchKernelGGL(k_wave_max, dim3(RED_BLOCKS, B), dim3(256), 0, st, HW, field, pmax);
    FGS_LAUNCH_CHECK("k_wave_max");
    hipLaunchKernelGGL(k_wave_output, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, st, HW, p.w.background[0],
                       p.w.background[1], p.w.background[2], field, dw, pmax, scal, out_rgb, out_depth,
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.seg_off), (uint32_t)p.base.tiles);
    FGS_LAUNCH_CHECK("k_wave_output");
    fgs_stage_end(ST_FIELD_FWD, st);
    return FGS_OK;
}

int fgs_wave_backward(const FgsWaveDims *dims, const float *cameras, const float *pos, const float *scale,
                      const float *quat, const float *color, const float *opacity, const float *phase,
                      void *saved, void *scratch, const float *g_rgb, const float *g_depth, float *g_pos,
                      float *g_scale, float *g_quat, float *g_color, float *g_opacity, float *g_phase,
                      void *stream) {
    WavePlan p;
    int rc = make_wave_plan(dims, &p);
    if (rc) return rc;
    const void *ptrs[] = {cameras, pos, scale, quat, color, opacity, phase, saved, scratch, g_rgb, g_depth,
                          g_pos, g_scale, g_quat, g_color, g_opacity, g_phase};
    if ((rc = check_ptrs(ptrs, 17, "fgs_wave_backward"))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *sv = reinterpret_cast<char *>(saved), *sc = reinterpret_cast<char *>(scratch);
    const int B = p.w.batch, W = p.w.width, H = p.w.height;
    const size_t HW = p.HW;
    float2 *field = reinterpret_cast<float2 *>(sv + p.v_field);
    float2 *dw = reinterpret_cast<float2 *>(sv + p.v_dw);
    float *scal = reinterpret_cast<float *>(sv + p.v_scal);
    float2 *gfield = reinterpret_cast<float2 *>(sc + p.c_gfield);
    float2 *gdw = reinterpret_cast<float2 *>(sc + p.c_gdw);
    float *rows = reinterpret_cast<float *>(sc + p.c_rows);
    fgs_stage_begin(ST_FIELD_BWD, st);
    const dim3 gpix((unsigned)((HW + 255) / 256), B);
    float2 *psum = reinterpret_cast<float2 *>(sc + p.c_part);
    hipLaunchKernelGGL(k_wave_output_bwd1, dim3(RED_BLOCKS, B), dim3(256), 0, st, HW, p.w.background[0],
                       p.w.background[1], p.w.background[2], field, scal, g_rgb, psum);
    FGS_LAUNCH_CHECK("k_wave_output_bwd1");
    hipLaunchKernelGGL(k_wave_output_bwd2, gpix, dim3(256), 0, st, HW, p.w.background[0], p.w.background[1],
                       p.w.background[2], field, dw, scal, psum, g_rgb, g_depth, gfield, gdw);
    FGS_LAUNCH_CHECK("k_wave_output_bwd2");
    fgs_stage_end(ST_FIELD_BWD, st);
    fgs_stage_begin(ST_SPLAT_BWD, st);
    const uint32_t grid = (uint32_t)p.base.L.seg_capacity;  // depth-segment units
    hipLaunchKernelGGL((k_asm_splat<true, true, 1>), dim3(grid), dim3(64), 0, st, (uint32_t)p.base.tiles,
                       (uint32_t)p.base.L.tiles_x, 1u, (uint32_t)W, (uint32_t)H, (uint32_t)p.base.L.dup_capacity,
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.tile_order),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.ranges),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.dup_ids),
                       reinterpret_cast<const float *>(sv + p.base.L.rec), reinterpret_cast<const float *>(sv + p.v_ccs),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.dup_off), gfield, rows, gdw,
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.counters),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.seg_off),
                       reinterpret_cast<const uint32_t *>(sv + p.base.L.seg_tile), (uint32_t)p.base.L.seg_len);
    FGS_LAUNCH_CHECK("k_wave_splat_bwd");
    fgs_stage_end(ST_SPLAT_BWD, st);
    fgs_stage_begin(ST_PROJECT_BWD, st);
    if ((rc = fgs_launch_asm_project_bwd(p.base, cameras, pos, scale, quat, color, phase, p.w.phase_channels, sv, rows,
                                         g_pos, g_scale, g_quat, g_color, g_opacity, g_phase, st, true)))
        return rc;
    fgs_stage_end(ST_PROJECT_BWD, st);
    return FGS_OK;
}

}  // extern "C"
