// CPU sanitizer run of the product's plan / layout arithmetic (fresnel_amd/csrc/fgs_plan.cpp, the host-only translation
// unit of libfgs_hip.so): compiled together with it by g++ -fsanitize=address,undefined (tests/test_sanitizers.py).
// Sweeps valid and invalid FgsDims and checks the invariants every kernel launch relies on: sections 256-byte
// aligned, in increasing order, inside total_bytes, capacities consistent, tuning resolved to a legal choice.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../fresnel_amd/csrc/fgs_plan.h"

static char g_err[512];
void fgs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { ++fails; fprintf(stderr, "FAIL %s:%d %s -- ", __FILE__, __LINE__, #c); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); }

static void check_plan(const FgsDims &d, int layers, bool ckpt) {
    FgsPlan p;
    const int rc = fgs_make_plan(&d, &p, layers, ckpt);
    if (rc != FGS_OK) { CHECK(rc == FGS_EINVAL && g_err[0], "rc=%d", rc); return; }
    const FgsSavedLayout &L = p.L;
    const size_t secs[] = {L.rec, L.depth_key, L.tile_count, L.order, L.dup_off, L.counters, L.ranges, L.tile_order,
                           L.dup_ids, L.pix_state, L.phase_ckpt};
    size_t prev = 0;
    for (size_t i = 0; i < sizeof(secs) / sizeof(secs[0]); ++i) {
        CHECK(secs[i] % 256 == 0, "section %zu unaligned", i);
        CHECK(secs[i] >= prev && secs[i] <= L.total_bytes, "section %zu out of order", i);
        prev = secs[i];
    }
    CHECK(L.seg_off <= L.seg_tile && L.seg_tile <= L.seg_ckpt && L.seg_ckpt <= L.total_bytes, "segments");
    // k_project's key statistics: 16 bytes per block of 256 Gaussians and image, between the layer ids and the segment tables
    CHECK(p.s_keybits % 256 == 0 && p.s_keybits >= p.s_layer && p.s_keybits >= L.phase_ckpt &&
          p.s_keybits + (size_t)d.batch * ((d.num_gaussians + 255) / 256) * 16 <= L.seg_off, "key statistics");
    CHECK(L.tile_w == 16 || L.tile_w == 32, "tile_w=%d", L.tile_w);
    CHECK(L.tiles_x == (d.width + L.tile_w - 1) / L.tile_w && L.tiles_y == (d.height + 15) / 16, "tile grid");
    CHECK(p.tiles == L.tiles_x * L.tiles_y, "tiles");
    CHECK(L.seg_len >= 64 && L.seg_len <= 512 && L.seg_len % 64 == 0, "seg_len=%d", L.seg_len);
    CHECK(L.dup_capacity == (size_t)d.batch * d.num_gaussians * p.tiles_per_gauss, "dup capacity");
    CHECK(L.dup_capacity < (1ull << 32), "dup capacity 2^32");
    if (!d.use_phase) CHECK(L.seg_capacity == L.dup_capacity / L.seg_len + (size_t)d.batch * layers * p.tiles, "ucap");
    CHECK((1ull << p.tile_key_bits) >= (size_t)d.batch * layers * p.tiles, "tile key bits");
    const size_t sc[] = {p.s_keys0, p.s_keys1, p.s_vals0, p.s_vals1, p.s_hist, p.s_bsum, p.s_grows, p.s_plane, p.s_rsum};
    prev = 0;
    for (size_t i = 0; i < sizeof(sc) / sizeof(sc[0]); ++i) {
        CHECK(sc[i] % 256 == 0 && sc[i] >= prev && sc[i] <= p.s_total, "scratch section %zu", i);
        prev = sc[i];
    }
    CHECK(p.s_keys1 - p.s_keys0 >= L.dup_capacity * 4 && p.s_keys1 - p.s_keys0 >= (size_t)d.batch * d.num_gaussians * 4, "sort buffers");
    // the tile tables borrow one sort buffer: 2 words + 64 buckets per launch-order group for every 1024 lists
    const size_t lists = (size_t)d.batch * layers * p.tiles, tblk = (lists + FGS_TILE_TABLE_TILES - 1) / FGS_TILE_TABLE_TILES;
    CHECK(p.tile_table_words == tblk * (2 + 64 * (size_t)p.order_groups) && p.tile_table_words <= p.sort_words, "tile tables");
    CHECK(p.s_keys1 - p.s_keys0 >= p.sort_words * 4 && p.s_vals0 - p.s_keys1 >= p.sort_words * 4 &&
          p.s_vals1 - p.s_vals0 >= p.sort_words * 4 && p.s_hist - p.s_vals1 >= p.sort_words * 4, "sort buffer size");
    CHECK(p.s_total - p.s_rsum >= (size_t)d.batch * d.num_gaussians * 48, "row sums");
    if (d.use_phase) CHECK(p.fwd_parts == 0 && p.tile_w == 16, "phase path split");
    if (p.tile_w == 32) CHECK(p.fwd_parts >= 1 && p.fwd_parts <= 8, "wide tiles parts=%d", p.fwd_parts);
    // the layout must not depend on anything but the dims: a second evaluation is identical
    FgsPlan q;
    CHECK(fgs_make_plan(&d, &q, layers, ckpt) == FGS_OK && memcmp(&p.L, &q.L, sizeof(p.L)) == 0 && p.s_total == q.s_total, "not a pure function");
}

int main() {
    static const int sizes[] = {1, 7, 16, 17, 63, 64, 96, 128, 145, 256, 500, 512, 1024, 2048, 4096, 32768};
    static const int ns[] = {1, 2, 63, 64, 65, 256, 8192, 32768, 262144};
    static const int bs[] = {1, 2, 3, 8, 16, 64};
    long n = 0;
    for (int wi = 0; wi < 16; ++wi) for (int hi = 0; hi < 16; ++hi) for (int ni = 0; ni < 9; ++ni) for (int bi = 0; bi < 6; ++bi) {
        FgsDims d;
        memset(&d, 0, sizeof(d));
        d.batch = bs[bi]; d.num_gaussians = ns[ni]; d.width = sizes[wi]; d.height = sizes[hi];
        d.max_radius = 64.0f; d.num_cameras = 1; d.phase_amplitude = 0.25f;
        check_plan(d, 1, true); ++n;
        d.use_phase = 1; check_plan(d, 1, true); d.use_phase = 0;
        d.saturation_skip = 1; check_plan(d, 1, true); d.saturation_skip = 0;
        check_plan(d, 16, false);  // the splat renderers' layered grids
        d.num_cameras = d.batch; d.max_radius = 1.0f + (float)(rnd() % 300);
        d.tile_w = (rnd() & 1) ? 32 : 16; d.seg_len = 64 * (int)(rnd() % 9); d.bin_mode = (int)(rnd() % 3);
        static const int fvs[] = {0, 1, 2, 4, 8, 16, -1, -2, -4};
        d.fwd_variant = fvs[rnd() % 9];
        check_plan(d, 1, true); ++n;
    }
    // hostile dims: every one must be rejected with FGS_EINVAL, none may trap
    for (int i = 0; i < 20000; ++i) {
        FgsDims d;
        uint32_t *w = reinterpret_cast<uint32_t *>(&d);
        for (size_t k = 0; k < sizeof(d) / 4; ++k) w[k] = (rnd() & 3) ? rnd() % 70000u : rnd() * 65536u + rnd();
        static const float radii[] = {64.0f, -1.0f, 1e30f, 0.0f, __builtin_nanf(""), __builtin_inff(), 1e-30f, 40000.0f};
        d.max_radius = radii[rnd() % 8];
        FgsPlan p;
        (void)fgs_make_plan(&d, &p, 1 + (int)(rnd() % 3), rnd() & 1);
    }
    // one corrupted field on otherwise valid dims (gets past the first validity test far more often)
    for (int i = 0; i < 200000; ++i) {
        FgsDims d;
        memset(&d, 0, sizeof(d));
        d.batch = 1 + (int)(rnd() % 64); d.num_gaussians = 1 + (int)(rnd() % 40000); d.width = 1 + (int)(rnd() % 2048);
        d.height = 1 + (int)(rnd() % 2048); d.max_radius = 64.0f; d.num_cameras = 1;
        static const float radii[] = {64.0f, -1.0f, 1e30f, 0.0f, __builtin_nanf(""), __builtin_inff(), 1e-30f, 40000.0f, 3e9f};
        static const int32_t ints[] = {0, -1, 1, 2147483647, (int32_t)0x80000000, 65536, 32768, 32769, 1 << 30};
        switch (rnd() % 10) {
            case 0: d.batch = ints[rnd() % 9]; break;
            case 1: d.num_gaussians = ints[rnd() % 9]; break;
            case 2: d.width = ints[rnd() % 9]; break;
            case 3: d.height = ints[rnd() % 9]; break;
            case 4: d.max_radius = radii[rnd() % 9]; break;
            case 5: d.num_cameras = ints[rnd() % 9]; break;
            case 6: d.seg_len = ints[rnd() % 9]; break;
            case 7: d.fwd_variant = ints[rnd() % 9]; break;
            case 8: d.bin_mode = ints[rnd() % 9]; break;
            default: d.tile_w = ints[rnd() % 9]; break;
        }
        FgsPlan p;
        const int rc = fgs_make_plan(&d, &p, 1 + (int)(rnd() % 3) * 15, rnd() & 1);
        CHECK(rc == FGS_OK || rc == FGS_EINVAL, "rc=%d", rc);
    }
    CHECK(fgs_make_plan(nullptr, nullptr) == FGS_EINVAL, "null dims");
    printf("plan_sanitize: %ld plans checked, %d failures\n", n, fails);
    return fails ? 1 : 0;
}
