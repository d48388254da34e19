/* Sanitizer driver of the CPU ORACLE (test infrastructure): `make asan` compiles fgs_oracle.c together with this file
 * under -fsanitize=address,undefined and runs every exported function over seeded scenes, including the edge cases
 * the parity tests use (no visible Gaussian, ragged frame sizes, radius cap, phase path, needles).  SURVEY section 5:
 * "CPU restatement under ASan/UBSan".  No expected values here -- those are tests/test_oracle_vs_golden.py's job;
 * this run must simply finish without a sanitizer report. */
#include "fgs_oracle.c"
#include <stdio.h>

static uint64_t s_rng = 88172645463325252ull;
static float urand(void) { s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17; return (float)((s_rng >> 11) & 0xFFFFFF) / 16777216.0f; }
static float nrand(void) { float s = 0; for (int i = 0; i < 12; ++i) s += urand(); return s - 6.0f; }
#define NEWF(n) ((float *)calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(float)))
#define NEWI(n) ((int32_t *)calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(int32_t)))

static int64_t run_scene(int32_t N, int32_t W, int32_t H, int kind, int use_phase)
{
    FgsOrCamera cam;
    memset(&cam, 0, sizeof(cam));
    for (int i = 0; i < 4; ++i) cam.view[5 * i] = 1.0f;
    cam.fx = cam.fy = 0.8f * (float)(W > H ? W : H); cam.cx = W / 2.0f; cam.cy = H / 2.0f;
    cam.width = W; cam.height = H; cam.near_ = 0.01f; cam.far_ = 100.0f;
    float *pos = NEWF(3 * N), *scl = NEWF(3 * N), *quat = NEWF(4 * N), *col = NEWF(3 * N), *opa = NEWF(N), *ph = NEWF(N);
    for (int32_t n = 0; n < N; ++n) {
        for (int k = 0; k < 3; ++k) { pos[3 * n + k] = 0.5f * nrand(); col[3 * n + k] = urand(); }
        pos[3 * n + 2] += kind == 1 ? 2.0f : -2.0f;                      /* kind 1: everything behind the camera */
        for (int k = 0; k < 3; ++k) scl[3 * n + k] = kind == 2 ? 0.5f + urand() : 0.01f + 0.12f * urand();  /* 2: radius cap */
        if (kind == 3) { scl[3 * n] = 0.003f; scl[3 * n + 1] = 1.5f * urand(); }   /* needles / discs */
        for (int k = 0; k < 4; ++k) quat[4 * n + k] = nrand();
        opa[n] = 1.3f * urand(); ph[n] = urand();
        if (kind == 4) pos[3 * n + 2] = -2.0f - 0.25f * (float)(n % 8);   /* depth ties */
    }
    float *cov = NEWF(4 * N), *mean = NEWF(2 * N), *dep = NEWF(N), *rad = NEWF(N), *conic = NEWF(3 * N);
    uint8_t *vis = (uint8_t *)calloc((size_t)N + 1, 1);
    int32_t *bbox = NEWI(4 * N), *order = NEWI(N), *vs = NEWI(N), V = 0;
    fgs_or_project(N, pos, scl, quat, &cam, 64.0f, cov, mean, dep, rad, vis, bbox, conic);
    fgs_or_depth_order(N, dep, vis, order, vs, &V);
    const int64_t P = fgs_or_count_pairs(V, vs, bbox);
    for (int pass = 0; pass < 2; ++pass) {
        const int32_t ts = pass ? (32 | (16 << 16)) : 16;
        const int32_t tw = pass ? 32 : 16, T = ((W + tw - 1) / tw) * ((H + 15) / 16);
        int64_t *ranges = (int64_t *)calloc((size_t)T + 1, sizeof(int64_t));
        const int64_t D = fgs_or_tile_lists(V, vs, bbox, W, H, ts, ranges, NULL);
        int32_t *ids = NEWI(D);
        if (fgs_or_tile_lists(V, vs, bbox, W, H, ts, ranges, ids) != D) { fprintf(stderr, "tile list count changed\n"); exit(1); }
        free(ids); free(ranges);
    }
    const size_t HW = (size_t)W * H;
    const float bg[3] = {0.1f, 0.2f, 0.3f};
    float *rgb = NEWF(3 * HW), *od = NEWF(HW), *state = NEWF(5 * HW), *pT = NEWF(P), *pPhi = NEWF(P);
    fgs_or_composite_fwd(V, vs, mean, conic, col, opa, dep, bbox, use_phase ? ph : NULL, 0.25f, W, H, bg, rgb, od, state,
                         pT, use_phase ? pPhi : NULL);
    float *gI = NEWF(3 * HW), *gD = NEWF(HW);
    for (size_t i = 0; i < 3 * HW; ++i) gI[i] = nrand();
    for (size_t i = 0; i < HW; ++i) gD[i] = 0.1f * nrand();
    float *gm = NEWF(2 * N), *gc = NEWF(3 * N), *gcol = NEWF(3 * N), *gop = NEWF(N), *gdep = NEWF(N), *gph = NEWF(N);
    fgs_or_composite_bwd(V, vs, mean, conic, col, opa, dep, bbox, use_phase ? ph : NULL, 0.25f, W, H, bg, state, pT,
                         use_phase ? pPhi : NULL, gI, gD, gm, gc, gcol, gop, gdep, use_phase ? gph : NULL);
    float *gpos = NEWF(3 * N), *gscl = NEWF(3 * N), *gquat = NEWF(4 * N);
    fgs_or_project_bwd(N, pos, scl, quat, &cam, vis, gm, gc, gdep, gpos, gscl, gquat);
    int64_t P2 = P;
    if (!use_phase) {
        P2 = fgs_or_render_fwd_bwd(N, pos, scl, quat, col, opa, &cam, 64.0f, bg, gI, gD, rgb, od, gpos, gscl, gquat, gcol, gop);
        if (P2 != P) { fprintf(stderr, "pair count differs between the staged and the fused entry: %lld vs %lld\n", (long long)P, (long long)P2); exit(1); }
    }
    free(pos); free(scl); free(quat); free(col); free(opa); free(ph); free(cov); free(mean); free(dep); free(rad); free(conic);
    free(vis); free(bbox); free(order); free(vs); free(rgb); free(od); free(state); free(pT); free(pPhi); free(gI); free(gD);
    free(gm); free(gc); free(gcol); free(gop); free(gdep); free(gph); free(gpos); free(gscl); free(gquat);
    return P;
}

int main(void)
{
    static const int32_t shapes[][3] = {{256, 128, 128}, {300, 96, 96}, {64, 64, 64}, {96, 160, 160}, {400, 96, 96},
                                        {17, 145, 66}, {1, 7, 5}, {0, 16, 16}, {1000, 33, 250}, {128, 1, 1}};
    int64_t total = 0;
    for (int kind = 0; kind < 5; ++kind)
        for (size_t s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s)
            for (int ph = 0; ph < 2; ++ph) total += run_scene(shapes[s][0], shapes[s][1], shapes[s][2], kind, ph);
    printf("oracle asan run: %lld pairs composited, no sanitizer report\n", (long long)total);
    return 0;
}
