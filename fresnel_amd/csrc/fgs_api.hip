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
// Experiment builds (python -m fresnel_amd.build --define X=1 --suffix _x, selected through FGS_LIB for same-box A/B runs) say so
// here: build.py hands every unit FGS_EXPERIMENT_BUILD and this one the list of the defines, the Python binding prints the string to
// stderr when it loads such a library, and the records of the sweeps / benches carry fgs_version() -- a library built with a
// timing-only switch can not pass for the product unnoticed.
#ifdef FGS_EXPERIMENT_BUILD
#ifndef FGS_BUILD_DEFINES
#define FGS_BUILD_DEFINES "(defines not recorded)"
#endif
const char *fgs_version(void) { return "fgs-hip 0.2 (gfx950) EXPERIMENT BUILD, NOT THE PRODUCT: " FGS_BUILD_DEFINES; }
#else
const char *fgs_version(void) { return "fgs-hip 0.2 (gfx950)"; }
#endif

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
    // (sort_mode bits 1-2 = 3: the depth sort's blocks hand their histograms to each other through scratch words that must be clear)
    const bool handoff = (p.d.sort_mode >> 1) == 3;
    if ((rc = fgs_launch_project(p, cameras, pos, scale, quat, color, opacity, sv, st, 0, 0.0f, 0.0f,
                                 handoff ? reinterpret_cast<uint32_t *>(sc + p.s_hist) : nullptr,
                                 (uint32_t)(FGS_SORT_HANDOFF_PASSES * (size_t)p.d.batch * 16 * 256)))) return rc;
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
