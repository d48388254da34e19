/*
 * fgs.h -- C ABI of the MI355X-native Gaussian-splatting rasterizer (libfgs_hip.so).
 *
 * The reference (CalebisGross/fresnel) has no FFI for this path: its boundary is the
 * Python call surface of scripts/models/differentiable_renderer.py ("DR"):
 *     TileBasedRenderer.__init__   DR:434-450
 *     TileBasedRenderer.forward    DR:489-499 -> (3,H,W) [, (H,W)]   DR:684-686
 *     Camera                       DR:24-52
 *     ASMWaveFieldRenderer.forward DR:1150-1161
 * This header is what a binding for that surface calls (see INTEGRATION.md for the
 * ctypes stub).  Conventions:
 *   - plain pointers and sizes only; every data pointer is DEVICE memory (HBM) unless
 *     marked host; all tensors contiguous fp32 in the reference's layouts, with a leading
 *     batch dimension B (one reference call == B = 1);
 *   - functions enqueue work on `stream` and return; they never allocate, never
 *     synchronise and are re-entrant per (stream, workspace);
 *   - return 0 on success, negative FGS_E* on error; fgs_last_error() gives the text.
 */
#ifndef FGS_H
#define FGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGS_OK 0
#define FGS_EINVAL (-1)    /* bad dims / null pointer */
#define FGS_ELAUNCH (-2)   /* HIP launch error */
#define FGS_EUNSUPPORTED (-3)

#define FGS_TILE 16        /* tile height in pixels, and the width unless FgsSavedLayout.tile_w says 32 */
#define FGS_SEG 128        /* largest depth segment: list entries per backward work unit; a call uses
                              FgsSavedLayout.seg_len (64 for small problems, else FGS_SEG)            */
#define FGS_TUNE_AUTO 0    /* FgsDims.seg_len / fwd_variant / bin_mode: let the library choose        */
#ifndef FGS_PHASE_CKPT
#define FGS_PHASE_CKPT 8   /* phase path: entries of a sub-tile's (compacted) list between (A, Phi) checkpoints */
#endif
#define FGS_CAMERA_FLOATS 24

/* Problem shape.  Mirrors TileBasedRenderer.__init__ (DR:434-450). */
typedef struct FgsDims {
    int32_t batch;          /* B images rendered by one call                       */
    int32_t num_gaussians;  /* N Gaussians per image                               */
    int32_t width, height;  /* image_width, image_height                 DR:445-446 */
    float max_radius;       /* radius cap in pixels (default 64)         DR:439,485 */
    float background[3];    /*                                            DR:447    */
    int32_t use_phase;      /* use_phase_blending && phases given        DR:629     */
    float phase_amplitude;  /*                                            DR:442    */
    int32_t num_cameras;    /* 1 (shared, TGD:1209-1223) or B                       */
    int32_t saturation_skip; /* 0 (default): every list entry is composited, like the reference (no early-out).
                                1: at every FGS_SEG-th list entry, 8x8 sub-tiles whose transmittance has fallen
                                below 2^-25 for all their pixels (accumulated alpha == 1.0f in fp32) stop being
                                composited, forward and backward; what is dropped is < 3e-8 * |colour| per pixel.
                                Blend path only (ignored with use_phase).                                      */
    /* Tuning overrides.  0 (FGS_TUNE_AUTO) everywhere = the measured-fastest choice for the launch size; the
     * settings only change how the work is split, never what is computed (results agree to fp32 rounding, the
     * integer stages bit for bit -- tests/test_hip_parity.py).  The library reads NO environment variables: a
     * forward and its backward agree because both derive the plan from the same FgsDims, and the forward also
     * records (seg_len, fwd_variant) in saved.counters[4..5], which the backward kernels read. */
    int32_t seg_len;        /* list entries per depth segment: 0 | a multiple of 64 up to 512 (saturation_skip: 128) */
    int32_t fwd_variant;    /* forward work split: 0 | 1, 2, 4, 8, 16 = depth-split forward with that many list parts
                               per tile (16 x 16 tiles; at most 8 on 32 x 16 tiles) | -1, -2, -4 = row-split forward with that many waves per tile.
                               Phase path: one work split (0 or 4).                                             */
    int32_t bin_mode;       /* tile binning: 0 | 1 = direct (column / row rank masks) | 2 = emit + stable radix sort */
    int32_t tile_w;         /* tile width in pixels: 0 | 16 | 32 (tiles are always 16 rows high).  0 = automatic:
                               32 on the blend path with the depth-split forward for frames >= 512 pixels wide
                               in calls of >= 3072 16 x 16 tiles, 16 elsewhere (FgsSavedLayout.tile_w tells)    */
    int32_t sort_mode;      /* depth sort.  Bit 0: 0 = radix passes over the 32 key bits | 1 = "zone keys" (BASELINE config 4,
                               --use_fresnel_zones: depths snapped to a few values): the keys are compressed to the bits that
                               vary over an image's visible Gaussians and only the passes those need do any work -- the same
                               order for ANY depths, faster when they vary in few bits.  NOT chosen automatically: what the
                               depths look like is known on the device only (fgs_sort.hip).
                               Bits 1-3 (work split, never the result): 0 = automatic -- images of <= 8192 Gaussians are sorted by
                               ONE launch, all passes in the LDS of one compute unit per image (round 5); larger ones by the
                               two-launch 8-bit passes of rounds 1-4 | 2 = one launch per pass, 11-bit digits, every block
                               recounting its image (<= 65 536 Gaussians) | 4 = the same with 8-bit digits | 6 = 8-bit digits,
                               the blocks' digit counts handed off between them instead of recounted (bounded wait) | 8 = the
                               two-launch passes for any size | 10 = as automatic.  Valid values: 0 ... 11.                 */
} FgsDims;

/* Camera record on the DEVICE: FGS_CAMERA_FLOATS floats per camera (Camera, DR:27-52):
 *   [0..15] view matrix row-major (world->camera), [16] fx, [17] fy, [18] cx, [19] cy,
 *   [20] near, [21] far, [22..23] unused. */

/* Byte offsets of the sections of the `saved` buffer (forward -> backward state and
 * the integer stages the parity tests inspect).  All sections 256-byte aligned. */
typedef struct FgsSavedLayout {
    size_t total_bytes;
    size_t rec;        /* float  [B][N][12]: u,v, conic(a, b+c, d), opacity, r,g,b, depth,
                                              bits(x0|x1<<16), bits(y0|y1<<16)            */
    size_t depth_key;  /* uint32 [B][N]: order-preserving depth bits, 0xFFFFFFFF = culled */
    size_t tile_count; /* uint32 [B][N]: tiles touched (0 = culled or empty bbox)         */
    size_t order;      /* uint32 [B][N]: Gaussian ids in canonical depth order (DR:527)   */
    size_t dup_off;    /* uint32 [B][N]: first duplicate slot of each Gaussian (emission
                                         order = image, depth rank, tile row, tile column)   */
    size_t counters;   /* uint32 [16]: [0] total duplicates D, [1] overflow flag, [2] depth-segment units U,
                                       [4] seg_len and [5] fwd_variant the forward ran with          */
    size_t ranges;     /* uint32 [B*T][2]: [start,end) into dup_ids per (image,tile)      */
    size_t tile_order; /* uint32 [B*T]: (image,tile) indices, longest lists first: the launch
                                         order of the composite kernels (scheduling only)    */
    size_t dup_ids;    /* uint32 [Dcap]: b*N+n per duplicate, sorted by (image,tile), depth
                                         order inside a tile                              */
    size_t pix_state;  /* float  [B][6][H][W]: C_r,C_g,C_b (pre-bg, pre-clamp), A, D, Phi */
    size_t phase_ckpt; /* float  [slots][8][64] (use_phase only): per-pixel (A, Phi) of a tile's four 8 x 8 sub-tiles (planes w and
                          4 + w) in front of every 8th entry THAT TOUCHES sub-tile w within a 64-entry block of the
                          tile's list; slot = start/8 + block offset/8 + group + tile                     */
    size_t dup_capacity; /* Dcap (elements, not bytes)                                    */
    int32_t tiles_x, tiles_y;
    /* Depth segments (non-phase path): a tile's list is cut into segments of FGS_SEG entries; each
     * segment is one work unit of the backward, so a launch is balanced however uneven the lists are. */
    size_t seg_off;    /* uint32 [B*T+1]: first unit of each (image,tile); [B*T] = number of units U
                                          (also counters[2])                                        */
    size_t seg_tile;   /* uint32 [Ucap]: (image,tile) of each unit; segment index = unit - seg_off    */
    size_t seg_ckpt;   /* float  [Ucap][5][4][64]: per-pixel C_r,C_g,C_b,A,D of the tile at the START
                                          of each unit with segment index >= 1 (written by the forward) */
    size_t seg_capacity; /* Ucap = Dcap / seg_len + B*T                                             */
    int32_t seg_len;     /* list entries per depth segment for these dims: 64 when B*N <= 200 000 (more,
                            shorter work units for launches that would not fill the chip), else 128   */
    int32_t tile_w;      /* tile width in pixels (16 | 32; tiles_x counts tiles of this width); a tile has
                            tile_w / 8 x 2 sub-tiles of 8 x 8 pixels, and seg_ckpt slots hold 5 x that many x 64 floats */
} FgsSavedLayout;

/* Sizes of the two caller-provided device buffers.  `saved` must stay untouched between
 * fgs_forward and the matching fgs_backward; `scratch` may be reused immediately. */
int fgs_workspace_bytes(const FgsDims *dims, size_t *saved_bytes, size_t *scratch_bytes);
int fgs_saved_layout(const FgsDims *dims, FgsSavedLayout *layout);

/* Replaces TileBasedRenderer.forward (DR:489-686) for B images at once.
 *   pos (B,N,3)  scale (B,N,3)  quat (B,N,4 wxyz, unnormalised ok)  color (B,N,3)
 *   opacity (B,N)  phase (B,N) or NULL
 *   out_rgb (B,3,H,W) clamped to [0,1]; out_depth (B,H,W). */
int fgs_forward(const FgsDims *dims, const float *cameras, const float *pos, const float *scale,
                const float *quat, const float *color, const float *opacity, const float *phase,
                float *out_rgb, float *out_depth, void *saved, void *scratch, void *stream);

/* Replaces autograd through DR:519-686: gradients of sum(out_rgb*g_rgb)+sum(out_depth*g_depth).
 * g_* outputs are fully overwritten (zeros for culled Gaussians).  g_phase may be NULL
 * when dims->use_phase == 0. */
int fgs_backward(const FgsDims *dims, const float *cameras, const float *pos, const float *scale,
                 const float *quat, const float *color, const float *opacity, const float *phase,
                 const void *saved, void *scratch, const float *g_rgb, const float *g_depth,
                 float *g_pos, float *g_scale, float *g_quat, float *g_color, float *g_opacity,
                 float *g_phase, void *stream);

/* Number of composited Gaussian-pixels of the last forward on this `saved` buffer
 * (SURVEY §8d unit of work): enqueues a reduction that writes one uint64 to
 * `out_pairs` (device). */
int fgs_count_pairs(const FgsDims *dims, const void *saved, uint64_t *out_pairs, void *stream);

/* ------------------------------------------------------------------------------------------
 * Angular-spectrum wave-field renderer: replaces ASMWaveFieldRenderer.forward (DR:1150-1344)
 * and AngularSpectrumPropagator (DR:929-1065).  Gaussians are splatted as complex amplitudes
 * onto num_planes depth planes (DR:1233-1283), every plane/channel is propagated to the focal
 * plane with the angular-spectrum transfer function H = exp(i 2 pi z sqrt(max(1/l^2-fx^2-fy^2,0)))
 * (DR:989-999) using batched hipFFT/rocFFT transforms, summed (one inverse FFT per channel, by
 * linearity), converted to intensity, normalised by the per-image maximum and composed with
 * the background (DR:1315-1332).
 *   phase       (B,N) or (B,N,3) radians (phase_channels = 1 | 3)        DR:1181, 1274-1275
 *   wavelengths (3,) DEVICE floats, shared by the batch                   DR:1160
 *   out_rgb     (B,3,H,W)
 * fgs_asm_backward writes gradients of sum(out_rgb*g_rgb) w.r.t. all Gaussian inputs, the
 * phases and the three wavelengths (dkz/dlambda is taken as 0 where 1/l^2-fx^2-fy^2 <= 0; the
 * reference's autograd returns NaN/inf when a frequency lands exactly on that boundary).
 * fgs_asm_backward CONSUMES `saved` (the plane fields / spectra in it are overwritten by their
 * gradients): one backward per forward.
 * The first call for a shape builds the hipFFT plans (host work); later calls only enqueue. */
typedef struct FgsAsmDims {
    int32_t batch, num_gaussians, width, height;
    float max_radius;        /* DR:1087 */
    float background[3];     /* DR:1086 */
    int32_t num_planes;      /* num_depth_planes, DR:1088 */
    float depth_near, depth_far; /* depth_range, DR:1089 */
    float focal_depth;       /* DR:1090 */
    double pixel_pitch;      /* DR:1091.  DOUBLE, like the Python float the reference hands to torch.fft.fftfreq(n, d): the frequency
                                grid is k * (float)(1.0 / (n * d)), and a pitch rounded to fp32 first moves that factor by an ulp
                                for some (n, d) -- 96 samples at 1/200 -- which the near-evanescent terms of dL/dlambda feel */
    int32_t phase_channels;  /* 1: phases (B,N); 3: phases (B,N,3) */
    int32_t num_cameras;     /* 1 or B */
    int32_t bin_mode;        /* list building, as FgsDims.bin_mode: 0 = automatic | 1 = direct (rank masks; the depth order is
                                grouped by plane and a (plane, tile) list is a rank range of the masks) | 2 = emit + stable
                                radix sort over (image, plane, tile) keys.  Same lists either way (tests)                 */
} FgsAsmDims;

int fgs_asm_workspace_bytes(const FgsAsmDims *dims, size_t *saved_bytes, size_t *scratch_bytes);
int fgs_asm_forward(const FgsAsmDims *dims, const float *cameras, const float *pos, const float *scale,
                    const float *quat, const float *color, const float *opacity, const float *phase,
                    const float *wavelengths, float *out_rgb, void *saved, void *scratch, void *stream);
int fgs_asm_backward(const FgsAsmDims *dims, const float *cameras, const float *pos, const float *scale,
                     const float *quat, const float *color, const float *opacity, const float *phase,
                     const float *wavelengths, void *saved, void *scratch, const float *g_rgb,
                     float *g_pos, float *g_scale, float *g_quat, float *g_color, float *g_opacity,
                     float *g_phase, float *g_wavelengths, void *stream);

/* ------------------------------------------------------------------------------------------
 * WaveFieldRenderer (DR:689-926; selected by --use_wave_rendering / --use_qsr, TGD:1891-1897):
 * order-independent complex amplitude accumulation U = sum a c exp(i phi) (DR:832-887), intensity
 * -> sqrt -> per-image max normalisation -> background where the total amplitude is low
 * (DR:893-914); depth map = sum(a depth) / (sum a + 1e-8) (DR:889-891, 924).
 *   phase (B,N) or (B,N,3) radians; out_rgb (B,3,H,W); out_depth (B,H,W). */
typedef struct FgsWaveDims {
    int32_t batch, num_gaussians, width, height;
    float max_radius;
    float background[3];
    int32_t phase_channels;  /* 1 | 3 */
    int32_t num_cameras;     /* 1 or B */
} FgsWaveDims;

int fgs_wave_workspace_bytes(const FgsWaveDims *dims, size_t *saved_bytes, size_t *scratch_bytes);
int fgs_wave_forward(const FgsWaveDims *dims, const float *cameras, const float *pos, const float *scale,
                     const float *quat, const float *color, const float *opacity, const float *phase,
                     float *out_rgb, float *out_depth, void *saved, void *scratch, void *stream);
int fgs_wave_backward(const FgsWaveDims *dims, const float *cameras, const float *pos, const float *scale,
                      const float *quat, const float *color, const float *opacity, const float *phase,
                      void *saved, void *scratch, const float *g_rgb, const float *g_depth, float *g_pos,
                      float *g_scale, float *g_quat, float *g_color, float *g_opacity, float *g_phase,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * Standalone angular-spectrum propagation: replaces AngularSpectrumPropagator.propagate (DR:1000-1065) and its
 * autograd.  field / out / g_* are (C, H, W) interleaved complex64 (the reference's (H, W, C) layout is permuted by
 * the binding); z = DEVICE scalar propagation distance; wavelengths = DEVICE (C,).  `spectrum` (C,H,W complex) receives
 * fft2(field) and is what the backward needs; scratch from fgs_asm_propagate_workspace_bytes.  band_limit: clamp
 * 1/l^2 - fx^2 - fy^2 at 0 (DR:993-994).  dL/dwavelength is taken as 0 where the clamp binds (see fgs_asm_backward). */
int fgs_asm_propagate_workspace_bytes(int32_t height, int32_t width, int32_t channels, size_t *scratch_bytes);
int fgs_asm_propagate_forward(int32_t height, int32_t width, int32_t channels, double pixel_pitch, int32_t band_limit,
                              const float *field, const float *z, const float *wavelengths, float *out,
                              float *spectrum, void *scratch, void *stream);
int fgs_asm_propagate_backward(int32_t height, int32_t width, int32_t channels, double pixel_pitch, int32_t band_limit,
                               const float *spectrum, const float *z, const float *wavelengths, const float *g_out,
                               float *g_field, float *g_z, float *g_wavelengths, void *scratch, void *stream);

/* ------------------------------------------------------------------------------------------
 * Spectral / stencil losses on the rendered batch.
 * mode 0 = FrequencyDomainLoss (TGD:428-522): mean over (B,C,H,W) of w (|fft2(rendered)| - |fft2(target)|)^2, w = 1
 *          below the radial frequency `cutoff`, `high_weight` above.
 * mode 1 = PhaseRetrievalLoss (TGD:342-425): the same with w = 1 on the fields sqrt(max(I, 1e-8)) exp(i phi),
 *          phi = (2 pi / wavelength) |depth - focal_depth|; depth (B,H,W), wavelength = DEVICE scalar.
 * loss / g_loss / g_wavelength are DEVICE scalars.  The backward CONSUMES `saved` (spectra -> their gradients).
 * g_target / g_depth may be NULL. */
typedef struct FgsSpectralDims {
    int32_t images, channels, height, width;
    int32_t mode;
    float cutoff, high_weight;   /* mode 0 */
    float focal_depth;           /* mode 1 */
    int32_t reserved;            /* must be 0 */
} FgsSpectralDims;
int fgs_spectral_workspace_bytes(const FgsSpectralDims *dims, size_t *saved_bytes, size_t *scratch_bytes);
int fgs_spectral_loss_forward(const FgsSpectralDims *dims, const float *rendered, const float *target,
                              const float *depth, const float *wavelength, float *loss, void *saved, void *scratch,
                              void *stream);
int fgs_spectral_loss_backward(const FgsSpectralDims *dims, const float *rendered, const float *target,
                               const float *depth, const float *wavelength, void *saved, void *scratch,
                               const float *g_loss, float *g_rendered, float *g_target, float *g_depth,
                               float *g_wavelength, void *stream);
/* wave_equation_loss (TGD:781-835): mean squared Helmholtz residual lap(U) + (2 pi / wavelength)^2 U of `images`
 * fields (H,W) with the periodic 5-point Laplacian on a grid of spacing pixel_spacing.  residual (images,H,W) is
 * written by the forward and read by the backward; scratch = fgs_reduction_scratch_bytes() bytes. */
int fgs_helmholtz_loss_forward(int32_t images, int32_t height, int32_t width, float wavelength, float pixel_spacing,
                               const float *field, float *loss, float *residual, void *scratch, void *stream);
int fgs_helmholtz_loss_backward(int32_t images, int32_t height, int32_t width, float wavelength, float pixel_spacing,
                                const float *residual, const float *g_loss, float *g_field, void *stream);
size_t fgs_reduction_scratch_bytes(void);

/* ------------------------------------------------------------------------------------------
 * Importance-subsampling hand-off between decoder and rasterizer (--stochastic_k; reference
 * scripts/training/train_gaussian_decoder.py:1160-1187): the n_out Gaussians whose indices torch.multinomial
 * drew (DEVICE int64 (n_out,), unique, shared by the batch) are gathered out of every (B, n_in, .) tensor into
 * (B, n_out, .) in one launch; the backward scatters the (B, n_out, .) gradients back and zero-fills the rest.
 *   phase_channels 0 (no phases) | 1 (B,N) | 3 (B,N,3). */
int fgs_gather_forward(int32_t batch, int32_t n_in, int32_t n_out, int32_t phase_channels, const int64_t *indices,
                       const float *pos, const float *scale, const float *quat, const float *color,
                       const float *opacity, const float *phase, float *o_pos, float *o_scale, float *o_quat,
                       float *o_color, float *o_opacity, float *o_phase, void *stream);
int fgs_gather_backward(int32_t batch, int32_t n_in, int32_t n_out, int32_t phase_channels, const int64_t *indices,
                        const float *g_o_pos, const float *g_o_scale, const float *g_o_quat, const float *g_o_color,
                        const float *g_o_opacity, const float *g_o_phase, float *g_pos, float *g_scale, float *g_quat,
                        float *g_color, float *g_opacity, float *g_phase, void *stream);

/* Per-stage hipEvent timers (profiling aid; SURVEY §5 "tracing").  When enabled, every stage
 * launched by the fgs_*_forward / fgs_*_backward entry points is bracketed by an event pair on the caller's
 * stream.  fgs_stage_timing_read synchronises on the recorded events, ADDS the elapsed milliseconds per
 * stage to ms[FGS_NUM_STAGES] and the number of launches to count[FGS_NUM_STAGES], and clears
 * the record.  Stage order: project, depth_sort, dup_emit, tile_sort, tile_ranges, composite_fwd,
 * composite_bwd, project_bwd, and for the splat renderers (ASM / wave field): splat_fwd, field_fwd (FFTs,
 * transfer function, plane sum, normalisation and output), field_bwd (their adjoints), splat_bwd.
 * enable: 0 = off, 1 = all stages, otherwise a mask with bit (stage + 1) per selected stage -- every event
 * pair costs a few microseconds of stream time, so a benchmark times only the kernel it reports.
 * (The only entry points that allocate or synchronise; never called by the product path.) */
#define FGS_NUM_STAGES 12
int fgs_stage_timing_enable(int enable);
int fgs_stage_timing_read(float *ms, int32_t *count);

const char *fgs_last_error(void);
const char *fgs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FGS_H */
