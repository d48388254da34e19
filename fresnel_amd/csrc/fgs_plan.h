// Plan / layout arithmetic of libfgs_hip.so: every buffer offset, capacity and work-split choice as a pure function of
// FgsDims.  HOST-ONLY C++ with no HIP include, so that the same translation unit (fgs_plan.cpp) is also compiled by g++
// under AddressSanitizer / UBSan on the CPU (tests/test_sanitizers.py; GPU sanitizers are not available on the pool).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/fgs.h"

#define FGS_REC_FLOATS 12
#define FGS_WAVE 64

// record field indices (saved.rec)
// gradient-row field indices (scratch.grows): one 12-float row per (tile, Gaussian) duplicate
enum { G_U = 0, G_V, G_CA, G_CBC, G_CD, G_OP, G_CR, G_CG, G_CB, G_DEPTH, G_PHASE, G_PAD1 };
#define FGS_GROW_FLOATS 12
#ifndef FGS_BLEND_ROW_FLOATS
#define FGS_BLEND_ROW_FLOATS 10  /* gradient rows of the (non-phase) blend backward: exactly its ten sums.  16 (build experiment):
                                    one 64-byte line per row, written whole by sixteen lanes -- no partial-sector writes, but 1.6x
                                    the bytes each way; measured in round 3, see DESIGN_LOG.md 10.3 */
#endif
#define FGS_BIN_G 256  /* depth ranks per block of the direct binning (fgs_bin.hip) */
#define FGS_TILE_TABLE_TILES 1024  /* lists per block of the tile-table kernels (fgs_bin.hip k_tile_pre / k_tile_post) */
enum { R_U = 0, R_V, R_CA, R_CBC, R_CD, R_OP, R_CR, R_CG, R_CB, R_DEPTH, R_BBX, R_BBY };

struct FgsPlan {
    FgsDims d;
    FgsSavedLayout L;
    int32_t layers;           // independent tile grids per image (ASM depth planes; 1 for TBR)
    size_t s_layer;           // saved: uint32 [B][N] layer of each Gaussian (layers > 1 only)
    size_t s_keybits;         // saved: uint32 [B][ceil(N / 256)][4]: per projection block, OR / AND of the visible depth keys and
                              // the visible / culled flags (k_project -> the depth sort's choice of radix passes, fgs_sort.hip)
    int32_t tile_w;           // tile width in pixels: 16, or 32 on the blend path (FgsDims.tile_w / automatic)
    int32_t tiles;            // tiles per image
    int32_t tiles_per_gauss;  // worst-case tiles touched by one Gaussian
    uint32_t tile_key_bits;   // bits of (image*T + tile)
    // resolved tuning (FgsDims.seg_len / fwd_variant / bin_mode with FGS_TUNE_AUTO replaced by the choice)
    int32_t fwd_parts;        // list parts of the depth-split forward; 0 = the row-split forward (k_composite_fwd)
    int32_t fwd_waves;        // waves per tile of the row-split forward (also the phase path)
    int32_t fwd_variant;      // the same choice in FgsDims.fwd_variant encoding (recorded in saved.counters[5])
    bool direct_binning;      // counting sort straight from the bboxes instead of emit + radix sort
    bool depth_ordered;       // the consumer composites in depth order (blend / phase paths).  false on the splat renderers (ASM, wave:
                              // order-independent sums, DR:1233-1283 / DR:689-926): no depth sort, the lists hold their Gaussians in
                              // index order -- the order the reference's own loop adds them in
    int32_t order_groups;     // XCD groups of the forward's launch order (fgs_bin.hip tile_group): 8 on the blend path's
                              // depth-split forward, 1 elsewhere
    // scratch layout (bytes)
    size_t sort_words;        // uint32 words of EACH of the four sort buffers below
    size_t tile_table_words;  // words the tile-table kernels need in the one sort buffer they borrow (<= sort_words)
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

#define FGS_BIN_MAX_TILES 4096  /* direct binning: tiles per image */
#define FGS_MASK_MAX_LINES 512  /* mask binning: tile columns + tile rows per image (LDS of k_mask_build: 16 KB) */
/* 64-bit rank words per mask line of the mask binning (fgs_bin.hip), padded to a multiple of 8 */
static inline uint32_t fgs_mask_words(uint32_t n) { return (((n + 63u) / 64u) + 7u) & ~7u; }
#define FGS_SORT_HANDOFF_PASSES 4
size_t fgs_radix_hist_bytes(uint32_t seg_capacity, uint32_t num_segs);
uint32_t fgs_radix_blocks_per_seg(uint32_t seg_capacity, uint32_t num_segs);
