// fgs_make_plan: see fgs_plan.h.  Host-only; no HIP types.
#include <string.h>
#include "fgs_plan.h"

uint32_t fgs_radix_blocks_per_seg(uint32_t seg_capacity, uint32_t num_segs) {
    // <= 1024 blocks per segment, >= 1024 keys per block (measured on the 6.9 M-key tile sort: 256 / 512 / 1024 /
    // 2048 / 4096 blocks -> 0.217 / 0.160 / 0.148 / 0.166 / 0.198 ms)
    uint32_t bps = (seg_capacity + 1023) / 1024;
    uint32_t cap = 1024;
    if (num_segs > 1) cap = 64;
    if (bps > cap) bps = cap;
    if (bps < 1) bps = 1;
    return bps;
}

size_t fgs_radix_hist_bytes(uint32_t seg_capacity, uint32_t num_segs) {
    const size_t bps = fgs_radix_blocks_per_seg(seg_capacity, num_segs);
    size_t words = (size_t)num_segs * 256 * bps + (size_t)num_segs * 256;
    // the hand-off form of the fused pass (FgsDims.sort_mode bits 1-2 = 3): four pass regions of [segments][<= 16 blocks][256] words
    const size_t handoff = (size_t)FGS_SORT_HANDOFF_PASSES * num_segs * 16 * 256;
    if (handoff > words) words = handoff;
    return words * sizeof(uint32_t);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int fgs_make_plan(const FgsDims *d, FgsPlan *p, int layers, bool segment_ckpt) {
    if (!d || !p) { fgs_set_error("null dims"); return FGS_EINVAL; }
    if (d->batch < 1 || d->batch > 65535 /* images are a grid dimension of the projection and the depth sort */ || d->num_gaussians < 1 || d->width < 1 || d->height < 1 || d->width > 32768 ||
        d->height > 32768 || !(d->max_radius > 0.0f) || !(d->max_radius <= 65536.0f) /* also rejects NaN / inf: the tile span
        below converts 2 * max_radius to int */ || (d->num_cameras != 1 && d->num_cameras != d->batch)) {
        fgs_set_error("invalid dims: B=%d N=%d W=%d H=%d max_radius=%g num_cameras=%d", d->batch,
                      d->num_gaussians, d->width, d->height, (double)d->max_radius, d->num_cameras);
        return FGS_EINVAL;
    }
    const int fv = d->fwd_variant, afv = fv < -16 || fv > 16 ? 3 /* invalid; INT_MIN has no negation */ : (fv < 0 ? -fv : fv);
    if (d->seg_len < 0 || d->seg_len > 512 || d->seg_len % 64 != 0 || (afv != 0 && afv != 1 && afv != 2 && afv != 4 && !(fv == 8 || fv == 16)) ||
        d->bin_mode < 0 || d->bin_mode > 2 || (d->tile_w != 0 && d->tile_w != 16 && d->tile_w != 32) || d->sort_mode < 0 || d->sort_mode > 11) {
        fgs_set_error("invalid tuning: seg_len=%d fwd_variant=%d bin_mode=%d tile_w=%d sort_mode=%d", d->seg_len, d->fwd_variant,
                      d->bin_mode, d->tile_w, d->sort_mode);
        return FGS_EINVAL;
    }
    const size_t B = d->batch, N = d->num_gaussians;
    if (B * N >= (1ull << 31)) { fgs_set_error("B*N too large"); return FGS_EINVAL; }
    memset(p, 0, sizeof(*p));
    p->d = *d;
    p->layers = layers;
    // Tile width.  32 x 16 tiles (eight 8 x 8 sub-tiles per lane) on the blend path with the depth-split forward: a Gaussian
    // touches ~0.6x as many tiles, so everything paid per (tile, Gaussian) duplicate -- LDS record reads, row / column
    // terms, the ten-sum reduction and its gradient row, the row-sum traffic, the lists -- is paid 0.6x as often.  The phase
    // path (one wave per sub-tile), the row-split forward (saturation_skip / fwd_variant < 0) and the splat renderers
    // (layers > 1 or no segment checkpoints) keep 16 x 16.
    const bool wide_ok = !d->use_phase && !d->saturation_skip && d->fwd_variant >= 0 && layers == 1 && segment_ckpt;
    if (d->tile_w == 32 && !wide_ok) {
        fgs_set_error("tile_w=32 needs the blend path with the depth-split forward");
        return FGS_EINVAL;
    }
    // automatic: wide tiles from 512-pixel-wide frames on, when the call has at least 3072 16 x 16 tiles (fewer do not fill
    // the chip and the finer tiles' parallelism wins).  What really decides is how many tiles a Gaussian touches, which the
    // dims do not say; measured on the benchmark scenes, 32 x 16 against 16 x 16 per step: 512^2 at 8 images -4 % (config 3:
    // backward -4 ... -6 %, row sums -29 %, forward equal) and -5 % decoder-like, at 4 images -2.8 %, at 3 images -2.3 %
    // (decoder-like: equal), at 2 / 1 images +1 / +4.5 %; 256^2 (config 2, Gaussians half as large in pixels) +3 %.
    const size_t tiles16 = B * (size_t)((d->width + 15) / 16) * (size_t)((d->height + 15) / 16);
    p->tile_w = d->tile_w ? d->tile_w : ((wide_ok && d->width >= 512 && tiles16 >= 3072) ? 32 : 16);
    const int tx = (d->width + p->tile_w - 1) / p->tile_w, ty = (d->height + FGS_TILE - 1) / FGS_TILE;
    p->tiles = tx * ty;
    // bbox width <= floor(2r)+2 pixels -> spans at most floor((2r+1)/tile)+2 tile columns
    int span = (int)((2.0 * (double)d->max_radius + 1.0) / p->tile_w) + 2;
    if (span > tx) span = tx;
    int spany = (int)((2.0 * (double)d->max_radius + 1.0) / FGS_TILE) + 2;
    if (spany > ty) spany = ty;
    p->tiles_per_gauss = span * spany;
    const size_t dcap = B * N * (size_t)p->tiles_per_gauss;
    if (dcap >= (1ull << 32) - 256) { fgs_set_error("duplicate capacity exceeds 2^32"); return FGS_EINVAL; }
    uint32_t bits = 0;
    while ((1ull << bits) < B * (size_t)layers * p->tiles) ++bits;
    p->tile_key_bits = bits;

    FgsSavedLayout &L = p->L;
    size_t o = 0;
    L.rec = o; o = align256(o + B * N * FGS_REC_FLOATS * 4);
    L.depth_key = o; o = align256(o + B * N * 4);
    L.tile_count = o; o = align256(o + B * N * 4);
    L.order = o; o = align256(o + B * N * 4);
    L.dup_off = o; o = align256(o + B * N * 4);
    L.counters = o; o = align256(o + 16 * 4);
    L.ranges = o; o = align256(o + B * layers * p->tiles * 2 * 4);
    L.tile_order = o; o = align256(o + B * layers * p->tiles * 4);
    L.dup_ids = o; o = align256(o + dcap * 4);
    L.pix_state = o; o = align256(o + B * 6 * (size_t)d->width * d->height * 4);
    L.phase_ckpt = o;
    if (d->use_phase) o = align256(o + (dcap / FGS_PHASE_CKPT + B * p->tiles + 2) * 8 * 64 * 4);
    p->s_layer = o;
    if (layers > 1) o = align256(o + B * N * 4);
    p->s_keybits = o; o = align256(o + B * ((N + 255) / 256) * 4 * 4);
    // ---- tuning: a pure function of the dims (no environment, so a forward and its backward always agree) ----
    // Forward work split.  Blend path: depth-split forward with 4 list parts per tile, 1 part once the launch has
    // enough tiles to fill the chip several times over (fwd ms, 8 images x 1024 tiles: row-split 0.642, 2 parts
    // 0.586, 4 parts 0.588; config 2 (4096 tiles): 0.169 / 0.167 / 0.134; 32 images: one wave per tile 2.215,
    // 1 part 2.173, 2 parts 2.27).  saturation_skip runs on the row-split forward; its waves per tile: two halve
    // the serial length of the longest lists, one wins with >= 24576 tiles, four for launches that cannot fill the
    // chip once.  Phase path: the recurrence is latency-bound (serial cos / divide chain per pixel), one wave per sub-tile.
    const uint32_t grid_tiles = (uint32_t)(B * p->tiles);
    if (afv > 4 && (d->use_phase || d->saturation_skip || layers != 1 || !segment_ckpt)) {
        fgs_set_error("fwd_variant=%d: 8 / 16 list parts exist on the blend path's depth-split forward only", fv);
        return FGS_EINVAL;
    }
    if (d->use_phase) {
        // one wave per 8 x 8 sub-tile, four per block (k_phase_fwd / k_phase_bwd): the only work split of this path
        if (afv != 0 && afv != 4) {
            fgs_set_error("fwd_variant=%d: the phase path has one work split (one wave per 8 x 8 sub-tile)", fv);
            return FGS_EINVAL;
        }
        p->fwd_parts = 0;
        p->fwd_waves = 4;
        p->fwd_variant = -4;
    } else if (d->saturation_skip || fv < 0) {
        p->fwd_parts = 0;
        p->fwd_waves = fv < 0 ? afv : (grid_tiles >= 24576u ? 1 : (grid_tiles <= 6144u ? 4 : 2));
        p->fwd_variant = -p->fwd_waves;
    } else {
        // few tiles: the launch is as long as its longest list, so more parts per tile (fwd ms at 4 / 8 / 16 parts, 16 x 16
        // tiles: config 3 at 1 image -- 1024 tiles -- 0.186 / 0.120 / 0.111, at 2 images 0.204 / 0.167 / 0.204, at 3 images
        // 0.243 / 0.250 / 0.31; config 2 at 2 images -- 512 tiles -- 0.076 / 0.054 / 0.046, at 8 images 0.083 / 0.076 / 0.103;
        // 32 x 16 tiles -- launches of >= 2048 of them, two waves per part -- stay at 4: 0.315 vs 0.344 with 8 at config 3, 4 images)
        p->fwd_parts = fv > 0 ? fv : (grid_tiles >= 24576u ? 1 : (p->tile_w != 16 ? 4 : (grid_tiles <= 1024u ? 16 : (grid_tiles <= 2048u ? 8 : 4))));
        if (p->tile_w == 32 && p->fwd_parts > 8) p->fwd_parts = 8;  // two waves per part there: 16 waves per block
        p->fwd_waves = p->fwd_parts;
        p->fwd_variant = p->fwd_parts;
    }
    // depth-segment length: shorter segments = more, shorter backward work units; pays off when the launch would
    // not fill the chip a few times over (config 2: -5 %, config 5: -3 %), costs 1 % at config 3's size.  The
    // row-split forward stages up to 128 records per chunk and needs 128.
    const bool row_split = !d->use_phase && p->fwd_parts == 0;
    if (row_split && d->seg_len != 0 && d->seg_len != FGS_SEG) {
        fgs_set_error("seg_len=%d is not available with the row-split forward (saturation_skip / fwd_variant < 0)", d->seg_len);
        return FGS_EINVAL;
    }
#ifndef FGS_SEG64_MAX_GAUSSIANS
#define FGS_SEG64_MAX_GAUSSIANS 200000  /* B * N up to which the shorter segments pay; re-measured at the end of round 2: config 2 bwd 0.264 (64) vs 0.281 ms (128), config 3 at 4 images 0.684 vs 0.690, at 8 images 1.30 vs 1.29, decoder-like 3.52 vs 3.37 */
#endif
    L.seg_len = d->seg_len ? d->seg_len : ((B * N <= FGS_SEG64_MAX_GAUSSIANS && !row_split) ? 64 : FGS_SEG);
    p->direct_binning = p->tiles <= FGS_BIN_MAX_TILES && tx + ty <= FGS_MASK_MAX_LINES && d->bin_mode != 2;
    if (d->bin_mode == 1 && !p->direct_binning) {
        fgs_set_error("bin_mode=1 (direct binning) needs <= %d tiles and <= %d tile columns + rows per image",
                      FGS_BIN_MAX_TILES, FGS_MASK_MAX_LINES);
        return FGS_EINVAL;
    }
    L.tile_w = p->tile_w;
#ifndef FGS_ORDER_GROUPS
#define FGS_ORDER_GROUPS 8
#endif
    p->order_groups = (p->fwd_parts > 0 && layers == 1 && segment_ckpt) ? FGS_ORDER_GROUPS : 1;
#ifdef FGS_SPLAT_ORDER_GROUPS
    // build experiment (round 4): the splat renderers' forward launch order grouped per XCD as well -- XCD g walks the g-th
    // eighth of the (image, plane, tile) keys longest-first (eight images: one image per XCD, all eight concurrently)
    if (!segment_ckpt && B * (size_t)layers * p->tiles >= 24576) p->order_groups = FGS_ORDER_GROUPS;
#endif
    p->depth_ordered = segment_ckpt;  // the callers without segment checkpoints are the splat renderers
    const size_t ucap = dcap / L.seg_len + B * layers * p->tiles;
    L.seg_off = o; L.seg_tile = o; L.seg_ckpt = o; L.seg_capacity = 0;
    if (!d->use_phase) {
        L.seg_capacity = ucap;
        L.seg_off = o; o = align256(o + (B * layers * p->tiles + 1) * 4);
        L.seg_tile = o; o = align256(o + ucap * 4);
        L.seg_ckpt = o;
        if (segment_ckpt && layers == 1) o = align256(o + ucap * 5 * 64 * (size_t)(p->tile_w / 4) * 4);  // 5 x sub-tiles x 64 floats per slot
    }
    L.total_bytes = o;
    L.dup_capacity = dcap;
    L.tiles_x = tx; L.tiles_y = ty;

    size_t nsort = dcap > B * N ? dcap : B * N;
    // the direct (mask) binning keeps the list lengths [B * tiles] in the second sort buffer and its
    // [B][tile columns + rows][rank words] 64-bit masks in the first
    const size_t bin_words = B * (size_t)layers * p->tiles;
    if (bin_words > nsort) nsort = bin_words;
    const size_t mask_words = p->direct_binning ? B * (size_t)(tx + ty) * fgs_mask_words((uint32_t)N) * 2 : 0;
    if (mask_words > nsort) nsort = mask_words;
    // the tile tables (k_tile_pre / k_tile_post) borrow ONE sort buffer: per block of FGS_TILE_TABLE_TILES lists a 64-bit sum
    // (two words) and 64 length buckets per launch-order group (with 8 groups a small call -- few Gaussians, or
    // 1025 ... 1027 lists -- needs more than the B * N words the buffer used to have)
    p->tile_table_words = ((bin_words + FGS_TILE_TABLE_TILES - 1) / FGS_TILE_TABLE_TILES) * (2 + 64 * (size_t)p->order_groups);
    if (p->tile_table_words > nsort) nsort = p->tile_table_words;
    p->sort_words = nsort;
    // block sums of the duplicate-offset scan: per image and block of FGS_BIN_G depth ranks (direct binning) or per
    // 256 flat elements (radix path) -- whichever is more
    const size_t nblk = B * ((N + FGS_BIN_G - 1) / FGS_BIN_G) + (B * N + 255) / 256 + 1;
    size_t hist = fgs_radix_hist_bytes((uint32_t)N, (uint32_t)B);
    const size_t hist2 = fgs_radix_hist_bytes((uint32_t)dcap, 1);
    if (hist2 > hist) hist = hist2;
    o = 0;
    p->s_keys0 = o; o = align256(o + nsort * 4);
    p->s_keys1 = o; o = align256(o + nsort * 4);
    p->s_vals0 = o; o = align256(o + nsort * 4);
    p->s_vals1 = o; o = align256(o + nsort * 4);
    p->s_hist = o; o = align256(o + hist);
    p->s_bsum = o; o = align256(o + nblk * 4);
    // gradient rows: one per duplicate; four (one per sub-tile wave) on the phase path
    p->s_grows = o; o = align256(o + dcap * 4 * (d->use_phase ? 4 * FGS_GROW_FLOATS
                                                             : (FGS_BLEND_ROW_FLOATS > FGS_GROW_FLOATS ? FGS_BLEND_ROW_FLOATS : FGS_GROW_FLOATS)));
    p->s_plane = o; o = align256(o + B * ((size_t)layers + 1) * 4);
    p->s_rsum = o; o = align256(o + B * N * 12 * 4);
    p->s_total = o;
    return FGS_OK;
}

