// Spectral / stencil operators that sit next to the rasterizer in the training step and the
// standalone angular-spectrum propagator:
//
//   fgs_asm_propagate_*     AngularSpectrumPropagator.propagate (DR:1000-1065): U(z) = ifft2(fft2(U0) H),
//                           H = exp(i 2 pi z sqrt(max(1/l^2 - fx^2 - fy^2, 0))), per channel wavelength; with the
//                           adjoint (dL/dfield, dL/dz, dL/dwavelength)
//   fgs_spectral_loss_*     PhaseRetrievalLoss (TGD:342-425) and FrequencyDomainLoss (TGD:428-522): one batched C2C
//                           transform of BOTH images' fields, then ONE fused pass  sum w (|F_r| - |F_t|)^2  (the
//                           reference makes two masked spectra, four abs and two mse passes), and the adjoint
//   fgs_helmholtz_loss_*    wave_equation_loss (TGD:781-835): periodic 5-point Laplacian + k^2 U, squared mean, one
//                           fused stencil pass each way (the reference: five rolls and six elementwise kernels)
//
// All HBM-bound streaming kernels around rocFFT: 8 B per complex sample per pass.  Reductions are deterministic
// (per-block partial sums in double, summed in block order by one block).  Compiled without fast-math: the phases
// reach hundreds of radians, sin / cos need the accurate range reduction.
#include <hipfft/hipfft.h>
#include "fgs_internal.h"

namespace {

constexpr int RT = 256;        // threads per block of the streaming kernels
constexpr int MAX_PART = 1024; // partial sums per reduction

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
unsigned part_blocks(size_t n) {
    size_t b = (n + RT - 1) / RT;
    return (unsigned)(b > MAX_PART ? MAX_PART : (b ? b : 1));
}

__device__ __forceinline__ float fftfreq(int k, int n, float inv_nd) {
    const int ks = (k < (n + 1) / 2) ? k : k - n;
    return (float)ks * inv_nd;
}

// block sum of one double per thread (fixed order: wave shuffles, then the four wave totals in order)
__device__ __forceinline__ double block_sum(double v) {
    __shared__ double ws[RT / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < RT / 64; ++w) t += ws[w];
    return t;
}

// out[k] = scale * sum of part[k][0..nblk) in block order (k < nout)
__global__ __launch_bounds__(RT) void k_final_sums(const double *__restrict__ part, unsigned nblk, unsigned stride,
                                                   unsigned nout, double scale, float *__restrict__ out) {
    for (unsigned k = 0; k < nout; ++k) {
        double v = 0.0;
        for (unsigned i = threadIdx.x; i < nblk; i += RT) v += part[(size_t)k * stride + i];
        const double t = block_sum(v);
        if (threadIdx.x == 0) out[k] = (float)(t * scale);
        __syncthreads();
    }
}

// ---- standalone propagator ------------------------------------------------------------------------------------
// spec <- spec * H / (H W)   (the 1/(HW) of the normalised inverse transform folded in)      DR:989-999, 1045-1047
__global__ __launch_bounds__(RT) void k_prop_apply(int W, int H, int C, float inv_ndx, float inv_ndy, int band_limit,
                                                   const float *__restrict__ zp, const float *__restrict__ wl,
                                                   const float2 *__restrict__ spec, float2 *__restrict__ out) {
    const size_t HW = (size_t)W * H, i = (size_t)blockIdx.x * RT + threadIdx.x;
    if (i >= HW * C) return;
    const int c = (int)(i / HW), kx = (int)(i % W), ky = (int)((i / W) % H);
    const float fx = fftfreq(kx, W, inv_ndx), fy = fftfreq(ky, H, inv_ndy), il = 1.0f / wl[c];
    float kz2 = fgs_kz2(il, fx, fy);
    if (band_limit) kz2 = kz2 < 0.0f ? 0.0f : kz2;
    const float theta = (6.28318530717958647692f * zp[0]) * sqrtf(kz2);  // NaN for evanescent waves without band limit, as torch
    float sn, cs;
    sincosf(theta, &sn, &cs);
    const float2 f = spec[i];
    const float s = 1.0f / (float)HW;
    out[i] = make_float2((f.x * cs - f.y * sn) * s, (f.x * sn + f.y * cs) * s);
}

// gP = fft(g_out) / (HW) is in `g`; F = fft(field) in `spec`:  g <- gP conj(H) (dL/dF);  partial sums of dL/dz and
// dL/dlambda_c from dL/dtheta = Im(gH conj(H)), gH = gP conj(F).
__global__ __launch_bounds__(RT) void k_prop_apply_bwd(int W, int H, int C, float inv_ndx, float inv_ndy, int band_limit,
                                                       const float *__restrict__ zp, const float *__restrict__ wl,
                                                       const float2 *__restrict__ spec, float2 *__restrict__ g,
                                                       double *__restrict__ part /*[1 + C][gridDim.x]*/) {
    const size_t HW = (size_t)W * H;
    double dz = 0.0, dl = 0.0;
    const int c = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * RT + threadIdx.x; i < HW; i += (size_t)gridDim.x * RT) {
        const int kx = (int)(i % W), ky = (int)(i / W);
        const float fx = fftfreq(kx, W, inv_ndx), fy = fftfreq(ky, H, inv_ndy), lam = wl[c], il = 1.0f / lam;
        const float raw = fgs_kz2(il, fx, fy);
        const float kz2 = (band_limit && raw < 0.0f) ? 0.0f : raw;
        const float kz = sqrtf(kz2), z = zp[0];
        float sn, cs;
        sincosf((6.28318530717958647692f * z) * kz, &sn, &cs);
        const size_t o = (size_t)c * HW + i;
        const float s = 1.0f / (float)HW;
        const float2 gp = make_float2(g[o].x * s, g[o].y * s), f = spec[o];
        // gH = gP conj(F); dtheta = Im(gH conj(H))
        const float ghx = gp.x * f.x + gp.y * f.y, ghy = gp.y * f.x - gp.x * f.y;
        const float dth = ghy * cs - ghx * sn;
        dz += (double)dth * 6.28318530717958647692 * kz;
        // dkz/dlambda = -1 / (lambda^3 kz) where the wave propagates; 0 where the clamp binds (torch: sqrt'(0) = inf
        // times clamp' = 0 gives NaN there; documented deviation, as in fgs_asm_backward)
        if (raw > 0.0f) dl += (double)dth * 6.28318530717958647692 * z * (-1.0 / ((double)lam * lam * lam * kz));
        g[o] = make_float2(gp.x * cs + gp.y * sn, gp.y * cs - gp.x * sn);  // gP conj(H)
    }
    const double tz = block_sum(dz);
    const double tl = block_sum(dl);
    if (threadIdx.x == 0) {
        part[(size_t)0 * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x] = tz;
        part[(size_t)(1 + c) * (gridDim.x * gridDim.y) + blockIdx.x] = tl;
    }
}

// ---- spectral losses --------------------------------------------------------------------------------------------
struct SpecPlan {
    FgsSpectralDims d;
    size_t n;            // B*C*H*W
    size_t v_spec;       // saved: float2 [2][B*C][H][W]  fields -> spectra (-> gradients in the backward)
    size_t v_total;
    size_t c_part;       // scratch: double [2][MAX_PART]
    size_t c_work;
    size_t c_total;
};

int make_spec_plan(const FgsSpectralDims *d, SpecPlan *p) {
    if (!d || d->images < 1 || d->channels < 1 || d->height < 1 || d->width < 1 || (d->mode != 0 && d->mode != 1) ||
        d->reserved != 0) {
        fgs_set_error("invalid spectral dims");
        return FGS_EINVAL;
    }
    p->d = *d;
    p->n = (size_t)d->images * d->channels * d->height * d->width;
    if (p->n >= (1ull << 31)) { fgs_set_error("spectral loss: batch too large"); return FGS_EINVAL; }
    p->v_spec = 0;
    p->v_total = align256(2 * p->n * sizeof(float2));
    size_t work = 0;
    const int rc = fgs_fft2_work_bytes(d->height, d->width, 2 * d->images * d->channels, &work);
    if (rc) return rc;
    p->c_part = 0;
    p->c_work = align256(2 * MAX_PART * sizeof(double));
    p->c_total = p->c_work + align256(work + 256);
    return FGS_OK;
}

// field = amplitude * exp(i phi):  mode 0 (FrequencyDomainLoss): (image, 0);  mode 1 (PhaseRetrievalLoss):
// sqrt(max(image, 1e-8)) * exp(i (2 pi / lambda) |depth - focal|)                                    TGD:405-416
__global__ __launch_bounds__(RT) void k_spec_pack(int mode, int C, size_t HW, size_t n, float focal,
                                                  const float *__restrict__ rendered, const float *__restrict__ target,
                                                  const float *__restrict__ depth, const float *__restrict__ wl,
                                                  float2 *__restrict__ spec) {
    const size_t i = (size_t)blockIdx.x * RT + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) {
        spec[i] = make_float2(rendered[i], 0.0f);
        spec[n + i] = make_float2(target[i], 0.0f);
        return;
    }
    const size_t b = i / (HW * C), px = i % HW;
    const float phi = (6.28318530717958647692f / wl[0]) * fabsf(depth[b * HW + px] - focal);
    float sn, cs;
    sincosf(phi, &sn, &cs);
    const float ar = sqrtf(fmaxf(rendered[i], 1e-8f)), at = sqrtf(fmaxf(target[i], 1e-8f));
    spec[i] = make_float2(ar * cs, ar * sn);
    spec[n + i] = make_float2(at * cs, at * sn);
}

__device__ __forceinline__ float spec_weight(int mode, int kx, int ky, int W, int H, float cutoff, float high_weight) {
    if (mode != 0) return 1.0f;
    const float u = fftfreq(kx, W, 1.0f / (float)W), v = fftfreq(ky, H, 1.0f / (float)H);
    return sqrtf(u * u + v * v) < cutoff ? 1.0f : high_weight;  // TGD:474-480, 509-521
}

__global__ __launch_bounds__(RT) void k_spec_reduce(int mode, int W, int H, size_t n, float cutoff, float high_weight,
                                                    const float2 *__restrict__ spec, double *__restrict__ part) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * RT + threadIdx.x; i < n; i += (size_t)gridDim.x * RT) {
        const float2 a = spec[i], b = spec[n + i];
        const float d = sqrtf(a.x * a.x + a.y * a.y) - sqrtf(b.x * b.x + b.y * b.y);
        acc += (double)(spec_weight(mode, (int)(i % W), (int)((i / W) % H), W, H, cutoff, high_weight) * d * d);
    }
    const double t = block_sum(acc);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// spectra -> dL/dspectra in place: dL/dF_r = (2 g / n) w (|F_r| - |F_t|) F_r / |F_r|, dL/dF_t = -(...) F_t / |F_t|
__global__ __launch_bounds__(RT) void k_spec_grad(int mode, int W, int H, size_t n, float cutoff, float high_weight,
                                                  const float *__restrict__ g_loss, float2 *__restrict__ spec) {
    const size_t i = (size_t)blockIdx.x * RT + threadIdx.x;
    if (i >= n) return;
    const float2 a = spec[i], b = spec[n + i];
    const float ma = sqrtf(a.x * a.x + a.y * a.y), mb = sqrtf(b.x * b.x + b.y * b.y);
    const float k = (2.0f * g_loss[0] / (float)n) * spec_weight(mode, (int)(i % W), (int)((i / W) % H), W, H, cutoff, high_weight) * (ma - mb);
    const float ka = ma > 0.0f ? k / ma : 0.0f, kb = mb > 0.0f ? -k / mb : 0.0f;  // |.|'(0) = 0, as torch
    spec[i] = make_float2(ka * a.x, ka * a.y);
    spec[n + i] = make_float2(kb * b.x, kb * b.y);
}

// field gradients (after the adjoint transform) -> image / depth gradients; thread = one pixel of one image
__global__ __launch_bounds__(RT) void k_spec_unpack(int mode, int B, int C, size_t HW, float focal,
                                                    const float *__restrict__ rendered, const float *__restrict__ target,
                                                    const float *__restrict__ depth, const float *__restrict__ wl,
                                                    const float2 *__restrict__ g, float *__restrict__ g_rendered,
                                                    float *__restrict__ g_target, float *__restrict__ g_depth,
                                                    double *__restrict__ part) {
    const size_t n = (size_t)B * C * HW;
    double dl = 0.0;
    for (size_t t = (size_t)blockIdx.x * RT + threadIdx.x; t < (size_t)B * HW; t += (size_t)gridDim.x * RT) {
        const size_t b = t / HW, px = t % HW;
        float sn = 0.0f, cs = 1.0f, gphi = 0.0f, pd = 0.0f, lam = 1.0f, dd = 0.0f;
        if (mode == 1) {
            lam = wl[0];
            dd = depth[b * HW + px] - focal;
            pd = fabsf(dd);
            sincosf((6.28318530717958647692f / lam) * pd, &sn, &cs);
        }
        for (int c = 0; c < C; ++c) {
            const size_t i = (b * C + c) * HW + px;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const float2 gf = g[f * n + i];
                float *dst = f ? g_target : g_rendered;
                if (mode == 0) {
                    if (dst) dst[i] = gf.x;
                    continue;
                }
                const float img = f ? target[i] : rendered[i];
                const float amp = sqrtf(fmaxf(img, 1e-8f));
                const float gamp = gf.x * cs + gf.y * sn;
                if (dst) dst[i] = img >= 1e-8f ? gamp * 0.5f / amp : 0.0f;  // clamp(min) passes the gradient on [1e-8, inf)
                gphi += amp * (gf.y * cs - gf.x * sn);
            }
        }
        if (mode == 1) {
            const float sg = dd > 0.0f ? 1.0f : (dd < 0.0f ? -1.0f : 0.0f);
            if (g_depth) g_depth[b * HW + px] = gphi * (6.28318530717958647692f / lam) * sg;
            dl += (double)gphi * (-6.28318530717958647692 / ((double)lam * lam)) * pd;
        }
    }
    const double tl = block_sum(dl);
    if (threadIdx.x == 0) part[blockIdx.x] = tl;
}

// ---- Helmholtz residual ---------------------------------------------------------------------------------------
// r = (U[y,x-1] + U[y,x+1] + U[y-1,x] + U[y+1,x] - 4 U) / h^2 + k^2 U, periodic                        TGD:820-831
__global__ __launch_bounds__(RT) void k_helmholtz(int W, int H, size_t n, float inv_h2, float k2,
                                                  const float *__restrict__ u, float *__restrict__ res,
                                                  double *__restrict__ part) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * RT + threadIdx.x; i < n; i += (size_t)gridDim.x * RT) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const size_t base = i - (size_t)y * W - x;
        const float c = u[i];
        const float ring = ((u[base + (size_t)y * W + (x ? x - 1 : W - 1)] + u[base + (size_t)y * W + (x + 1 < W ? x + 1 : 0)]) +
                            u[base + (size_t)(y ? y - 1 : H - 1) * W + x]) + u[base + (size_t)(y + 1 < H ? y + 1 : 0) * W + x];
        const float r = (ring - 4.0f * c) * inv_h2 + k2 * c;
        if (res) res[i] = r;
        acc += (double)r * r;
    }
    const double t = block_sum(acc);
    if (part && threadIdx.x == 0) part[blockIdx.x] = t;
}

// dL/dU = (2 g / n) * L r with the same (self-adjoint) stencil applied to the residual
__global__ __launch_bounds__(RT) void k_helmholtz_bwd(int W, int H, size_t n, float inv_h2, float k2,
                                                      const float *__restrict__ res, const float *__restrict__ g_loss,
                                                      float *__restrict__ g_u) {
    const size_t i = (size_t)blockIdx.x * RT + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const size_t base = i - (size_t)y * W - x;
    const float c = res[i];
    const float ring = ((res[base + (size_t)y * W + (x ? x - 1 : W - 1)] + res[base + (size_t)y * W + (x + 1 < W ? x + 1 : 0)]) +
                        res[base + (size_t)(y ? y - 1 : H - 1) * W + x]) + res[base + (size_t)(y + 1 < H ? y + 1 : 0) * W + x];
    g_u[i] = (2.0f * g_loss[0] / (float)n) * ((ring - 4.0f * c) * inv_h2 + k2 * c);
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------------------------
int fgs_asm_propagate_workspace_bytes(int32_t height, int32_t width, int32_t channels, size_t *scratch_bytes) {
    if (height < 1 || width < 1 || channels < 1) { fgs_set_error("fgs_asm_propagate: invalid dims"); return FGS_EINVAL; }
    size_t work = 0;
    const int rc = fgs_fft2_work_bytes(height, width, channels, &work);
    if (rc) return rc;
    if (scratch_bytes) *scratch_bytes = align256((size_t)(1 + channels) * MAX_PART * sizeof(double)) + align256(work + 256);
    return FGS_OK;
}

int fgs_asm_propagate_forward(int32_t height, int32_t width, int32_t channels, double pixel_pitch, int32_t band_limit,
                              const float *field, const float *z, const float *wavelengths, float *out,
                              float *spectrum, void *scratch, void *stream) {
    size_t sb;
    int rc = fgs_asm_propagate_workspace_bytes(height, width, channels, &sb);
    if (rc) return rc;
    if (!field || !z || !wavelengths || !out || !spectrum || !scratch || !(pixel_pitch > 0.0)) {
        fgs_set_error("fgs_asm_propagate_forward: null pointer / bad pitch");
        return FGS_EINVAL;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t n = (size_t)channels * height * width;
    char *work = reinterpret_cast<char *>(scratch) + align256((size_t)(1 + channels) * MAX_PART * sizeof(double));
    hipError_t e = hipMemcpyAsync(spectrum, field, n * sizeof(float2), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { fgs_set_error("propagate copy: %s", hipGetErrorString(e)); return FGS_ELAUNCH; }
    float2 *spec = reinterpret_cast<float2 *>(spectrum), *o = reinterpret_cast<float2 *>(out);
    if ((rc = fgs_fft2_exec(height, width, channels, spec, HIPFFT_FORWARD, work, st))) return rc;
    const float inv_ndx = (float)(1.0 / ((double)width * pixel_pitch));
    const float inv_ndy = (float)(1.0 / ((double)height * pixel_pitch));
    hipLaunchKernelGGL(k_prop_apply, dim3((unsigned)((n + RT - 1) / RT)), dim3(RT), 0, st, width, height, channels, inv_ndx,
                       inv_ndy, band_limit, z, wavelengths, spec, o);
    FGS_LAUNCH_CHECK("k_prop_apply");
    return fgs_fft2_exec(height, width, channels, o, HIPFFT_BACKWARD, work, st);
}

int fgs_asm_propagate_backward(int32_t height, int32_t width, int32_t channels, double pixel_pitch, int32_t band_limit,
                               const float *spectrum, const float *z, const float *wavelengths, const float *g_out,
                               float *g_field, float *g_z, float *g_wavelengths, void *scratch, void *stream) {
    size_t sb;
    int rc = fgs_asm_propagate_workspace_bytes(height, width, channels, &sb);
    if (rc) return rc;
    if (!spectrum || !z || !wavelengths || !g_out || !g_field || !g_z || !g_wavelengths || !scratch) {
        fgs_set_error("fgs_asm_propagate_backward: null pointer argument");
        return FGS_EINVAL;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t HW = (size_t)height * width, n = HW * channels;
    double *part = reinterpret_cast<double *>(scratch);
    char *work = reinterpret_cast<char *>(scratch) + align256((size_t)(1 + channels) * MAX_PART * sizeof(double));
    hipError_t e = hipMemcpyAsync(g_field, g_out, n * sizeof(float2), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { fgs_set_error("propagate copy: %s", hipGetErrorString(e)); return FGS_ELAUNCH; }
    float2 *g = reinterpret_cast<float2 *>(g_field);
    // adjoint of the normalised inverse transform: forward transform / (HW) (the scale is applied in the kernel)
    if ((rc = fgs_fft2_exec(height, width, channels, g, HIPFFT_FORWARD, work, st))) return rc;
    unsigned gx = part_blocks(HW);
    if ((size_t)gx * channels > MAX_PART) gx = MAX_PART / channels ? MAX_PART / channels : 1;
    const float inv_ndx = (float)(1.0 / ((double)width * pixel_pitch));
    const float inv_ndy = (float)(1.0 / ((double)height * pixel_pitch));
    hipLaunchKernelGGL(k_prop_apply_bwd, dim3(gx, channels), dim3(RT), 0, st, width, height, channels, inv_ndx, inv_ndy,
                       band_limit, z, wavelengths, reinterpret_cast<const float2 *>(spectrum), g, part);
    FGS_LAUNCH_CHECK("k_prop_apply_bwd");
    // adjoint of the forward transform: the unnormalised inverse
    if ((rc = fgs_fft2_exec(height, width, channels, g, HIPFFT_BACKWARD, work, st))) return rc;
    const unsigned stride = gx * channels;
    hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(RT), 0, st, part, stride, stride, 1u, 1.0, g_z);
    FGS_LAUNCH_CHECK("k_final_sums");
    hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(RT), 0, st, part + stride, gx, stride, (unsigned)channels, 1.0,
                       g_wavelengths);
    FGS_LAUNCH_CHECK("k_final_sums");
    return FGS_OK;
}

// ------------------------------------------------------------------------------------------------------------------
int fgs_spectral_workspace_bytes(const FgsSpectralDims *dims, size_t *saved_bytes, size_t *scratch_bytes) {
    SpecPlan p;
    const int rc = make_spec_plan(dims, &p);
    if (rc) return rc;
    if (saved_bytes) *saved_bytes = p.v_total;
    if (scratch_bytes) *scratch_bytes = p.c_total;
    return FGS_OK;
}

int fgs_spectral_loss_forward(const FgsSpectralDims *dims, const float *rendered, const float *target,
                              const float *depth, const float *wavelength, float *loss, void *saved, void *scratch,
                              void *stream) {
    SpecPlan p;
    int rc = make_spec_plan(dims, &p);
    if (rc) return rc;
    if (!rendered || !target || !loss || !saved || !scratch || (p.d.mode == 1 && (!depth || !wavelength))) {
        fgs_set_error("fgs_spectral_loss_forward: null pointer argument");
        return FGS_EINVAL;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int B = p.d.images, C = p.d.channels, H = p.d.height, W = p.d.width;
    const size_t HW = (size_t)H * W;
    float2 *spec = reinterpret_cast<float2 *>(saved);
    double *part = reinterpret_cast<double *>(scratch);
    char *work = reinterpret_cast<char *>(scratch) + p.c_work;
    hipLaunchKernelGGL(k_spec_pack, dim3((unsigned)((p.n + RT - 1) / RT)), dim3(RT), 0, st, p.d.mode, C, HW, p.n,
                       p.d.focal_depth, rendered, target, depth, wavelength, spec);
    FGS_LAUNCH_CHECK("k_spec_pack");
    if ((rc = fgs_fft2_exec(H, W, 2 * B * C, spec, HIPFFT_FORWARD, work, st))) return rc;
    const unsigned nb = part_blocks(p.n);
    hipLaunchKernelGGL(k_spec_reduce, dim3(nb), dim3(RT), 0, st, p.d.mode, W, H, p.n, p.d.cutoff, p.d.high_weight, spec, part);
    FGS_LAUNCH_CHECK("k_spec_reduce");
    hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(RT), 0, st, part, nb, nb, 1u, 1.0 / (double)p.n, loss);
    FGS_LAUNCH_CHECK("k_final_sums");
    return FGS_OK;
}

int fgs_spectral_loss_backward(const FgsSpectralDims *dims, const float *rendered, const float *target,
                               const float *depth, const float *wavelength, void *saved, void *scratch,
                               const float *g_loss, float *g_rendered, float *g_target, float *g_depth,
                               float *g_wavelength, void *stream) {
    SpecPlan p;
    int rc = make_spec_plan(dims, &p);
    if (rc) return rc;
    if (!rendered || !target || !saved || !scratch || !g_loss || !g_rendered ||
        (p.d.mode == 1 && (!depth || !wavelength || !g_wavelength))) {
        fgs_set_error("fgs_spectral_loss_backward: null pointer argument");
        return FGS_EINVAL;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int B = p.d.images, C = p.d.channels, H = p.d.height, W = p.d.width;
    const size_t HW = (size_t)H * W;
    float2 *spec = reinterpret_cast<float2 *>(saved);
    double *part = reinterpret_cast<double *>(scratch);
    char *work = reinterpret_cast<char *>(scratch) + p.c_work;
    hipLaunchKernelGGL(k_spec_grad, dim3((unsigned)((p.n + RT - 1) / RT)), dim3(RT), 0, st, p.d.mode, W, H, p.n, p.d.cutoff,
                       p.d.high_weight, g_loss, spec);
    FGS_LAUNCH_CHECK("k_spec_grad");
    // adjoint of the unnormalised forward transform: the unnormalised inverse
    if ((rc = fgs_fft2_exec(H, W, 2 * B * C, spec, HIPFFT_BACKWARD, work, st))) return rc;
    const unsigned nb = part_blocks((size_t)B * HW);
    hipLaunchKernelGGL(k_spec_unpack, dim3(nb), dim3(RT), 0, st, p.d.mode, B, C, HW, p.d.focal_depth, rendered, target,
                       depth, wavelength, spec, g_rendered, g_target, g_depth, part);
    FGS_LAUNCH_CHECK("k_spec_unpack");
    if (p.d.mode == 1) {
        hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(RT), 0, st, part, nb, nb, 1u, 1.0, g_wavelength);
        FGS_LAUNCH_CHECK("k_final_sums");
    }
    return FGS_OK;
}

// ------------------------------------------------------------------------------------------------------------------
int fgs_helmholtz_loss_forward(int32_t images, int32_t height, int32_t width, float wavelength, float pixel_spacing,
                               const float *field, float *loss, float *residual, void *scratch, void *stream) {
    if (images < 1 || height < 1 || width < 1 || !(wavelength > 0.0f) || !(pixel_spacing > 0.0f) || !field || !loss ||
        !residual || !scratch) {
        fgs_set_error("fgs_helmholtz_loss_forward: invalid argument");
        return FGS_EINVAL;
    }
    const size_t n = (size_t)images * height * width;
    const float k = 6.28318530717958647692f / wavelength;
    const unsigned nb = part_blocks(n);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    double *part = reinterpret_cast<double *>(scratch);
    hipLaunchKernelGGL(k_helmholtz, dim3(nb), dim3(RT), 0, st, width, height, n, 1.0f / (pixel_spacing * pixel_spacing),
                       k * k, field, residual, part);
    FGS_LAUNCH_CHECK("k_helmholtz");
    hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(RT), 0, st, part, nb, nb, 1u, 1.0 / (double)n, loss);
    FGS_LAUNCH_CHECK("k_final_sums");
    return FGS_OK;
}

int fgs_helmholtz_loss_backward(int32_t images, int32_t height, int32_t width, float wavelength, float pixel_spacing,
                                const float *residual, const float *g_loss, float *g_field, void *stream) {
    if (images < 1 || height < 1 || width < 1 || !(wavelength > 0.0f) || !(pixel_spacing > 0.0f) || !residual || !g_loss ||
        !g_field) {
        fgs_set_error("fgs_helmholtz_loss_backward: invalid argument");
        return FGS_EINVAL;
    }
    const size_t n = (size_t)images * height * width;
    const float k = 6.28318530717958647692f / wavelength;
    hipLaunchKernelGGL(k_helmholtz_bwd, dim3((unsigned)((n + RT - 1) / RT)), dim3(RT), 0, reinterpret_cast<hipStream_t>(stream),
                       width, height, n, 1.0f / (pixel_spacing * pixel_spacing), k * k, residual, g_loss, g_field);
    FGS_LAUNCH_CHECK("k_helmholtz_bwd");
    return FGS_OK;
}

size_t fgs_reduction_scratch_bytes(void) { return (size_t)MAX_PART * sizeof(double); }

}  // extern "C"
