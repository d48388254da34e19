// Internal declarations shared by the HIP translation units of libfgs_hip.so.
// gfx950 (MI355X, CDNA4) only: wave64, 160 KiB LDS/CU, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fgs.h"

#define FGS_REC_FLOATS 12
#define FGS_WAVE 64

// record field indices (saved.rec)
// gradient-row field indices (scratch.grows): one 12-float row per (tile, Gaussian) duplicate
enum { G_U = 0, G_V, G_CA, G_CBC, G_CD, G_OP, G_CR, G_CG, G_CB, G_DEPTH, G_PHASE, G_PAD1 };
#define FGS_GROW_FLOATS 12
#define FGS_BLEND_ROW_FLOATS 10  /* gradient rows of the (non-phase) blend backward: exactly its ten sums */
#define FGS_BIN_G 256  /* depth ranks per block of the direct binning (fgs_bin.hip) */
enum { R_U = 0, R_V, R_CA, R_CBC, R_CD, R_OP, R_CR, R_CG, R_CB, R_DEPTH, R_BBX, R_BBY };

struct FgsPlan {
    FgsDims d;
    FgsSavedLayout L;
    int32_t layers;           // independent tile grids per image (ASM depth planes; 1 for TBR)
    size_t s_layer;           // saved: uint32 [B][N] layer of each Gaussian (layers > 1 only)
    int32_t tile_w;           // tile width in pixels: 16, or 32 on the blend path (FgsDims.tile_w / automatic)
    int32_t tiles;            // tiles per image
    int32_t tiles_per_gauss;  // worst-case tiles touched by one Gaussian
    uint32_t tile_key_bits;   // bits of (image*T + tile)
    // resolved tuning (FgsDims.seg_len / fwd_variant / bin_mode with FGS_TUNE_AUTO replaced by the choice)
    int32_t fwd_parts;        // list parts of the depth-split forward; 0 = the row-split forward (k_composite_fwd)
    int32_t fwd_waves;        // waves per tile of the row-split forward (also the phase path)
    int32_t fwd_variant;      // the same choice in FgsDims.fwd_variant encoding (recorded in saved.counters[5])
    bool direct_binning;      // counting sort straight from the bboxes instead of emit + radix sort
    // scratch layout (bytes)
    size_t s_total;
    size_t s_keys0, s_keys1;  // uint32 [max(B*N, Dcap)] radix ping/pong keys
    size_t s_vals0, s_vals1;  // uint32 [max(B*N, Dcap)] radix ping/pong payloads (vals of the final pass land in saved.dup_ids)
    size_t s_hist;            // uint32 radix histograms
    size_t s_bsum;            // uint32 block sums for the duplicate-offset scan
    size_t s_grows;           // float [Dcap][12]: per-duplicate gradient rows (composite bwd -> reduce)
    size_t s_plane;           // uint32 [B][layers + 1]: first depth rank of every layer (layered direct binning)
    size_t s_rsum;            // float [B*N][12]: per-Gaussian totals of the blend path's rows (k_row_sum -> k_project_bwd)
};

// `segment_ckpt`: reserve the per-segment forward checkpoints of the tile-based compositing path (the splat
// renderers cut their lists into the same depth segments but carry no state between them)
int fgs_make_plan(const FgsDims *dims, FgsPlan *plan, int layers = 1, bool segment_ckpt = true);
void fgs_set_error(const char *fmt, ...);

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
#define FGS_BIN_MAX_TILES 4096  /* direct binning: tiles per image */
#define FGS_MASK_MAX_LINES 512  /* mask binning: tile columns + tile rows per image (LDS of k_mask_build: 16 KB) */
/* 64-bit rank words per mask line of the mask binning (fgs_bin.hip), padded to a multiple of 8 */
static inline uint32_t fgs_mask_words(uint32_t n) { return (((n + 63u) / 64u) + 7u) & ~7u; }
void fgs_stage_begin(int stage, hipStream_t st);  // no-ops unless fgs_stage_timing_enable(1)
void fgs_stage_end(int stage, hipStream_t st);

// ---- stage launchers (each enqueues on `st`, returns FGS_OK / FGS_ELAUNCH) ----
// plane_* != 0 selects the ASM variant: additionally writes saved.layer = nearest depth plane
// (DR:1136-1148) of each Gaussian
int fgs_launch_project(const FgsPlan &p, const float *cams, const float *pos, const float *scale,
                       const float *quat, const float *color, const float *opacity, char *saved,
                       hipStream_t st, int num_planes = 0, float plane_near = 0.0f, float plane_far = 0.0f);
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
                          uint32_t index_payload_mod = 0);
size_t fgs_radix_hist_bytes(uint32_t seg_capacity, uint32_t num_segs);

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
