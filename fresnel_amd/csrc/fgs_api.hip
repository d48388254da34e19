// C ABI of libfgs_hip.so (include/fgs.h): plan/layout computation and stage orchestration.
// Nothing here allocates device memory or synchronises; every call enqueues on the caller's
// stream (graph-capturable, Guideline 9 of the CDNA HIP guide).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "fgs_internal.h"

static thread_local char g_err[512] = "";

void fgs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-stage event timers -------------------------------------------------------------
#include <mutex>
#include <stdlib.h>
#include <vector>
namespace {
struct StageRec { int stage; hipEvent_t a, b; };
std::mutex g_tm;
unsigned g_timing = 0;  // bit (stage) set: that stage is bracketed by events
std::vector<StageRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open[FGS_NUM_STAGES];

hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void fgs_stage_begin(int stage, hipStream_t st) {
    if (!((g_timing >> stage) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_tm);
    hipEvent_t e = take_event();
    if (!e) return;
    (void)hipEventRecord(e, st);
    g_open[stage] = e;
}

void fgs_stage_end(int stage, hipStream_t st) {
    if (!((g_timing >> stage) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_tm);
    if (!g_open[stage]) return;
    hipEvent_t e = take_event();
    if (!e) return;
    (void)hipEventRecord(e, st);
    g_recs.push_back({stage, g_open[stage], e});
    g_open[stage] = nullptr;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int fgs_make_plan(const FgsDims *d, FgsPlan *p, int layers, bool segment_ckpt) {
    if (!d || !p) { fgs_set_error("null dims"); return FGS_EINVAL; }
    if (d->batch < 1 || d->num_gaussians < 1 || d->width < 1 || d->height < 1 || d->width > 32768 ||
        d->height > 32768 || !(d->max_radius > 0.0f) || (d->num_cameras != 1 && d->num_cameras != d->batch)) {
        fgs_set_error("invalid dims: B=%d N=%d W=%d H=%d max_radius=%g num_cameras=%d", d->batch,
                      d->num_gaussians, d->width, d->height, (double)d->max_radius, d->num_cameras);
        return FGS_EINVAL;
    }
    const int fv = d->fwd_variant, afv = fv < 0 ? -fv : fv;
    if (d->seg_len < 0 || d->seg_len > 512 || d->seg_len % 64 != 0 || (afv != 0 && afv != 1 && afv != 2 && afv != 4 && !(fv == 8 || fv == 16)) ||
        d->bin_mode < 0 || d->bin_mode > 2 || (d->tile_w != 0 && d->tile_w != 16 && d->tile_w != 32)) {
        fgs_set_error("invalid tuning: seg_len=%d fwd_variant=%d bin_mode=%d tile_w=%d", d->seg_len, d->fwd_variant,
                      d->bin_mode, d->tile_w);
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
    // ---- tuning: a pure function of the dims (no environment, so a forward and its backward always agree) ----
    // Forward work split.  Blend path: depth-split forward with 4 list parts per tile, 1 part once the launch has
    // enough tiles to fill the chip several times over (fwd ms, 8 images x 1024 tiles: row-split 0.642, 2 parts
    // 0.586, 4 parts 0.588; config 2 (4096 tiles): 0.169 / 0.167 / 0.134; 32 images: one wave per tile 2.215,
    // 1 part 2.173, 2 parts 2.27).  saturation_skip runs on the row-split forward; its waves per tile: two halve
    // the serial length of the longest lists, one wins with >= 24576 tiles, four for launches that cannot fill the
    // chip once.  Phase path: the recurrence is latency-bound (serial cos / divide chain per pixel), four waves.
    const uint32_t grid_tiles = (uint32_t)(B * p->tiles);
    if (afv > 4 && (d->use_phase || d->saturation_skip || layers != 1 || !segment_ckpt)) {
        fgs_set_error("fwd_variant=%d: 8 / 16 list parts exist on the blend path's depth-split forward only", fv);
        return FGS_EINVAL;
    }
    if (d->use_phase) {
        p->fwd_parts = 0;
        p->fwd_waves = afv ? afv : 4;
        p->fwd_variant = -p->fwd_waves;
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
    p->s_grows = o; o = align256(o + dcap * FGS_GROW_FLOATS * 4 * (d->use_phase ? 4 : 1));
    p->s_plane = o; o = align256(o + B * ((size_t)layers + 1) * 4);
    p->s_rsum = o; o = align256(o + B * N * 12 * 4);
    p->s_total = o;
    return FGS_OK;
}

extern "C" {

const char *fgs_last_error(void) { return g_err; }

int fgs_stage_timing_enable(int enable) {
    std::lock_guard<std::mutex> lk(g_tm);
    // 0: off; 1: every stage; otherwise bit (stage + 1) selects individual stages
    g_timing = enable == 1 ? ~0u : ((unsigned)enable >> 1);
    return FGS_OK;
}

int fgs_stage_timing_read(float *ms, int32_t *count) {
    std::lock_guard<std::mutex> lk(g_tm);
    for (const StageRec &r : g_recs) {
        float t = 0.0f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            if (ms) ms[r.stage] += t;
            if (count) count[r.stage] += 1;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return FGS_OK;
}
const char *fgs_version(void) { return "fgs-hip 0.1 (gfx950)"; }

int fgs_workspace_bytes(const FgsDims *dims, size_t *saved_bytes, size_t *scratch_bytes) {
    FgsPlan p;
    const int rc = fgs_make_plan(dims, &p);
    if (rc) return rc;
    if (saved_bytes) *saved_bytes = p.L.total_bytes;
    if (scratch_bytes) *scratch_bytes = p.s_total;
    return FGS_OK;
}

int fgs_saved_layout(const FgsDims *dims, FgsSavedLayout *layout) {
    FgsPlan p;
    const int rc = fgs_make_plan(dims, &p);
    if (rc) return rc;
    if (!layout) { fgs_set_error("null layout"); return FGS_EINVAL; }
    *layout = p.L;
    return FGS_OK;
}

int fgs_forward(const FgsDims *dims, const float *cameras, const float *pos, const float *scale,
                const float *quat, const float *color, const float *opacity, const float *phase,
                float *out_rgb, float *out_depth, void *saved, void *scratch, void *stream) {
    FgsPlan p;
    int rc = fgs_make_plan(dims, &p);
    if (rc) return rc;
    if (!cameras || !pos || !scale || !quat || !color || !opacity || !out_rgb || !out_depth || !saved || !scratch) {
        fgs_set_error("fgs_forward: null pointer argument");
        return FGS_EINVAL;
    }
    if (p.d.use_phase && !phase) { fgs_set_error("fgs_forward: use_phase set but phase is NULL"); return FGS_EINVAL; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *sv = reinterpret_cast<char *>(saved), *sc = reinterpret_cast<char *>(scratch);
    fgs_stage_begin(ST_PROJECT, st);
    if ((rc = fgs_launch_project(p, cameras, pos, scale, quat, color, opacity, sv, st))) return rc;
    fgs_stage_end(ST_PROJECT, st);
    if ((rc = fgs_launch_binning(p, sv, sc, st))) return rc;
    fgs_stage_begin(ST_COMPOSITE_FWD, st);
    if ((rc = fgs_launch_composite_fwd(p, phase, sv, out_rgb, out_depth, st))) return rc;
    fgs_stage_end(ST_COMPOSITE_FWD, st);
    return FGS_OK;
}

int fgs_backward(const FgsDims *dims, const float *cameras, const float *pos, const float *scale,
                 const float *quat, const float *color, const float *opacity, const float *phase,
                 const void *saved, void *scratch, const float *g_rgb, const float *g_depth,
                 float *g_pos, float *g_scale, float *g_quat, float *g_color, float *g_opacity,
                 float *g_phase, void *stream) {
    FgsPlan p;
    int rc = fgs_make_plan(dims, &p);
    if (rc) return rc;
    if (!cameras || !pos || !scale || !quat || !color || !opacity || !saved || !scratch || !g_rgb || !g_depth ||
        !g_pos || !g_scale || !g_quat || !g_color || !g_opacity) {
        fgs_set_error("fgs_backward: null pointer argument");
        return FGS_EINVAL;
    }
    if (p.d.use_phase && (!phase || !g_phase)) {
        fgs_set_error("fgs_backward: use_phase set but phase/g_phase is NULL");
        return FGS_EINVAL;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const char *sv = reinterpret_cast<const char *>(saved);
    char *sc = reinterpret_cast<char *>(scratch);
    fgs_stage_begin(ST_COMPOSITE_BWD, st);
    if ((rc = fgs_launch_composite_bwd(p, phase, sv, sc, g_rgb, g_depth, g_phase, st))) return rc;
    fgs_stage_end(ST_COMPOSITE_BWD, st);
    fgs_stage_begin(ST_PROJECT_BWD, st);
    if ((rc = fgs_launch_project_bwd(p, cameras, pos, scale, quat, sv,
                                     reinterpret_cast<const float *>(sc + p.s_grows), g_pos, g_scale, g_quat,
                                     g_color, g_opacity, p.d.use_phase ? g_phase : nullptr, st,
                                     reinterpret_cast<float *>(sc + p.s_rsum))))
        return rc;
    fgs_stage_end(ST_PROJECT_BWD, st);
    return FGS_OK;
}

int fgs_count_pairs(const FgsDims *dims, const void *saved, uint64_t *out_pairs, void *stream) {
    FgsPlan p;
    const int rc = fgs_make_plan(dims, &p);
    if (rc) return rc;
    if (!saved || !out_pairs) { fgs_set_error("fgs_count_pairs: null pointer"); return FGS_EINVAL; }
    return fgs_launch_count_pairs(p, reinterpret_cast<const char *>(saved), out_pairs,
                                  reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
