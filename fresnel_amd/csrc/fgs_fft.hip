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
    if (r == HIPFFT_SUCCESS)
        r = hipfftMakePlanMany(pl->handle, 2, n, nullptr, 1, H * W, nullptr, 1, H * W, HIPFFT_C2C, batch, &pl->work);
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
