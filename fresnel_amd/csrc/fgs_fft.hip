// hipFFT (rocFFT) plan cache shared by the angular-spectrum renderer, the standalone propagator and the spectral
// losses: batched 2-D C2C transforms, in place, caller-provided work areas.
#include <hipfft/hipfft.h>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include "fgs_internal.h"

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
