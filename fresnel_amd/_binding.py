"""ctypes binding of libfgs_hip.so (include/fgs.h).  No torch types cross this boundary:
only raw device pointers, sizes and the HIP stream handle.

The product path FAILS LOUDLY when the HIP library is missing -- there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FGS_LIB", os.path.join(_HERE, "_lib", "libfgs_hip.so"))  # FGS_LIB: A/B builds

FGS_CAMERA_FLOATS = 24
FGS_TILE = 16

# every symbol include/fgs.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "fgs_workspace_bytes", "fgs_saved_layout", "fgs_forward", "fgs_backward", "fgs_count_pairs",
    "fgs_last_error", "fgs_version", "fgs_stage_timing_enable", "fgs_stage_timing_read",
    "fgs_asm_workspace_bytes", "fgs_asm_forward", "fgs_asm_backward",
    "fgs_wave_workspace_bytes", "fgs_wave_forward", "fgs_wave_backward",
    "fgs_gather_forward", "fgs_gather_backward",
    "fgs_asm_propagate_workspace_bytes", "fgs_asm_propagate_forward", "fgs_asm_propagate_backward",
    "fgs_spectral_workspace_bytes", "fgs_spectral_loss_forward", "fgs_spectral_loss_backward",
    "fgs_helmholtz_loss_forward", "fgs_helmholtz_loss_backward", "fgs_reduction_scratch_bytes",
]

STAGES = ["project", "depth_sort", "dup_emit", "tile_sort", "tile_ranges", "composite_fwd",
          "composite_bwd", "project_bwd", "splat_fwd", "field_fwd", "field_bwd", "splat_bwd"]


class FgsDims(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("num_gaussians", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("max_radius", ctypes.c_float), ("background", ctypes.c_float * 3),
                ("use_phase", ctypes.c_int32), ("phase_amplitude", ctypes.c_float),
                ("num_cameras", ctypes.c_int32), ("saturation_skip", ctypes.c_int32),
                ("seg_len", ctypes.c_int32), ("fwd_variant", ctypes.c_int32), ("bin_mode", ctypes.c_int32),
                ("tile_w", ctypes.c_int32), ("sort_mode", ctypes.c_int32)]


class FgsSavedLayout(ctypes.Structure):
    _fields_ = [("total_bytes", ctypes.c_size_t), ("rec", ctypes.c_size_t),
                ("depth_key", ctypes.c_size_t), ("tile_count", ctypes.c_size_t),
                ("order", ctypes.c_size_t), ("dup_off", ctypes.c_size_t), ("counters", ctypes.c_size_t),
                ("ranges", ctypes.c_size_t), ("tile_order", ctypes.c_size_t), ("dup_ids", ctypes.c_size_t),
                ("pix_state", ctypes.c_size_t), ("phase_ckpt", ctypes.c_size_t),
                ("dup_capacity", ctypes.c_size_t),
                ("tiles_x", ctypes.c_int32), ("tiles_y", ctypes.c_int32),
                ("seg_off", ctypes.c_size_t), ("seg_tile", ctypes.c_size_t), ("seg_ckpt", ctypes.c_size_t),
                ("seg_capacity", ctypes.c_size_t), ("seg_len", ctypes.c_int32), ("tile_w", ctypes.c_int32)]


class FgsAsmDims(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("num_gaussians", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("max_radius", ctypes.c_float), ("background", ctypes.c_float * 3),
                ("num_planes", ctypes.c_int32), ("depth_near", ctypes.c_float),
                ("depth_far", ctypes.c_float), ("focal_depth", ctypes.c_float),
                ("pixel_pitch", ctypes.c_double), ("phase_channels", ctypes.c_int32),
                ("num_cameras", ctypes.c_int32), ("bin_mode", ctypes.c_int32)]


class FgsWaveDims(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("num_gaussians", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("max_radius", ctypes.c_float), ("background", ctypes.c_float * 3),
                ("phase_channels", ctypes.c_int32), ("num_cameras", ctypes.c_int32)]


class FgsSpectralDims(ctypes.Structure):
    _fields_ = [("images", ctypes.c_int32), ("channels", ctypes.c_int32), ("height", ctypes.c_int32),
                ("width", ctypes.c_int32), ("mode", ctypes.c_int32), ("cutoff", ctypes.c_float),
                ("high_weight", ctypes.c_float), ("focal_depth", ctypes.c_float), ("reserved", ctypes.c_int32)]


class FgsError(RuntimeError):
    pass


_lib = None


def load():
    """Load libfgs_hip.so; raise (never fall back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FgsError(
            f"{LIB_PATH} not found: build it with `python -m fresnel_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP rasterizer.")
    # torch first: it carries the process's HIP runtime (libamdhip64.so.7); loading ours before
    # it would bring up a second runtime from /opt/rocm that cannot see torch's device context.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, cp = ctypes.c_void_p, ctypes.POINTER
    lib.fgs_last_error.restype = ctypes.c_char_p
    lib.fgs_version.restype = ctypes.c_char_p
    if b"EXPERIMENT" in lib.fgs_version():  # an A/B build selected through FGS_LIB: say so, once, where nobody can miss it
        import sys
        sys.stderr.write(f"fresnel_amd: loaded {LIB_PATH}: {lib.fgs_version().decode()}\n")
    lib.fgs_workspace_bytes.argtypes = [cp(FgsDims), cp(ctypes.c_size_t), cp(ctypes.c_size_t)]
    lib.fgs_saved_layout.argtypes = [cp(FgsDims), cp(FgsSavedLayout)]
    lib.fgs_forward.argtypes = [cp(FgsDims)] + [vp] * 12
    lib.fgs_backward.argtypes = [cp(FgsDims)] + [vp] * 18
    lib.fgs_count_pairs.argtypes = [cp(FgsDims), vp, vp, vp]
    lib.fgs_stage_timing_enable.argtypes = [ctypes.c_int]
    lib.fgs_stage_timing_read.argtypes = [cp(ctypes.c_float), cp(ctypes.c_int32)]
    lib.fgs_stage_timing_enable.restype = ctypes.c_int
    lib.fgs_stage_timing_read.restype = ctypes.c_int
    lib.fgs_asm_workspace_bytes.argtypes = [cp(FgsAsmDims), cp(ctypes.c_size_t), cp(ctypes.c_size_t)]
    lib.fgs_asm_forward.argtypes = [cp(FgsAsmDims)] + [vp] * 12
    lib.fgs_asm_backward.argtypes = [cp(FgsAsmDims)] + [vp] * 19
    lib.fgs_wave_workspace_bytes.argtypes = [cp(FgsWaveDims), cp(ctypes.c_size_t), cp(ctypes.c_size_t)]
    lib.fgs_wave_forward.argtypes = [cp(FgsWaveDims)] + [vp] * 12
    lib.fgs_wave_backward.argtypes = [cp(FgsWaveDims)] + [vp] * 18
    i32 = ctypes.c_int32
    lib.fgs_gather_forward.argtypes = [i32, i32, i32, i32] + [vp] * 14
    lib.fgs_gather_backward.argtypes = [i32, i32, i32, i32] + [vp] * 14
    lib.fgs_gather_forward.restype = lib.fgs_gather_backward.restype = ctypes.c_int
    f32 = ctypes.c_float
    lib.fgs_asm_propagate_workspace_bytes.argtypes = [i32, i32, i32, cp(ctypes.c_size_t)]
    lib.fgs_asm_propagate_forward.argtypes = [i32, i32, i32, ctypes.c_double, i32] + [vp] * 7   # (the pitch is a double: fgs.h)
    lib.fgs_asm_propagate_backward.argtypes = [i32, i32, i32, ctypes.c_double, i32] + [vp] * 9
    lib.fgs_spectral_workspace_bytes.argtypes = [cp(FgsSpectralDims), cp(ctypes.c_size_t), cp(ctypes.c_size_t)]
    lib.fgs_spectral_loss_forward.argtypes = [cp(FgsSpectralDims)] + [vp] * 8
    lib.fgs_spectral_loss_backward.argtypes = [cp(FgsSpectralDims)] + [vp] * 12
    lib.fgs_helmholtz_loss_forward.argtypes = [i32, i32, i32, f32, f32] + [vp] * 5
    lib.fgs_helmholtz_loss_backward.argtypes = [i32, i32, i32, f32, f32] + [vp] * 4
    lib.fgs_reduction_scratch_bytes.argtypes = []
    lib.fgs_reduction_scratch_bytes.restype = ctypes.c_size_t
    for fn in (lib.fgs_asm_propagate_workspace_bytes, lib.fgs_asm_propagate_forward, lib.fgs_asm_propagate_backward,
               lib.fgs_spectral_workspace_bytes, lib.fgs_spectral_loss_forward, lib.fgs_spectral_loss_backward,
               lib.fgs_helmholtz_loss_forward, lib.fgs_helmholtz_loss_backward):
        fn.restype = ctypes.c_int
    for fn in (lib.fgs_asm_workspace_bytes, lib.fgs_asm_forward, lib.fgs_asm_backward,
               lib.fgs_wave_workspace_bytes, lib.fgs_wave_forward, lib.fgs_wave_backward):
        fn.restype = ctypes.c_int
    for fn in (lib.fgs_workspace_bytes, lib.fgs_saved_layout, lib.fgs_forward, lib.fgs_backward,
               lib.fgs_count_pairs):
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


def version():
    """fgs_version() of the loaded library (experiment builds list their defines there)."""
    return load().fgs_version().decode()


def check(rc, what):
    if rc != 0:
        raise FgsError(f"{what} failed (rc={rc}): {load().fgs_last_error().decode()}")


def make_dims(batch, num_gaussians, width, height, max_radius=64.0, background=(0.0, 0.0, 0.0),
              use_phase=False, phase_amplitude=0.25, num_cameras=1, saturation_skip=False, tuning=None):
    """`tuning`: optional dict of FgsDims overrides {seg_len, fwd_variant, bin_mode, tile_w, sort_mode} (0 / absent = automatic)."""
    d = FgsDims()
    d.batch, d.num_gaussians, d.width, d.height = int(batch), int(num_gaussians), int(width), int(height)
    d.max_radius = float(max_radius)
    for i in range(3):
        d.background[i] = float(background[i])
    d.use_phase = 1 if use_phase else 0
    d.phase_amplitude = float(phase_amplitude)
    d.num_cameras = int(num_cameras)
    d.saturation_skip = 1 if saturation_skip else 0
    for k, v in (tuning or {}).items():
        if k not in ("seg_len", "fwd_variant", "bin_mode", "tile_w", "sort_mode"):
            raise FgsError(f"unknown tuning field {k!r}")
        setattr(d, k, int(v))
    return d


def workspace_bytes(dims):
    s, c = ctypes.c_size_t(0), ctypes.c_size_t(0)
    check(load().fgs_workspace_bytes(ctypes.byref(dims), ctypes.byref(s), ctypes.byref(c)),
          "fgs_workspace_bytes")
    return s.value, c.value


def saved_layout(dims):
    L = FgsSavedLayout()
    check(load().fgs_saved_layout(ctypes.byref(dims), ctypes.byref(L)), "fgs_saved_layout")
    return L


def stage_timing_enable(on=True, stages=None):
    """on=True: every stage; stages=[names]: only those (each event pair costs a few microseconds of stream time)."""
    mask = 1 if on else 0
    if on and stages is not None:
        mask = 0
        for name in stages:
            mask |= 1 << (STAGES.index(name) + 1)
    check(load().fgs_stage_timing_enable(mask), "fgs_stage_timing_enable")


def stage_timing_read():
    """-> {stage: (total_ms, launches)} accumulated since the previous read."""
    n = len(STAGES)
    ms = (ctypes.c_float * n)()
    cnt = (ctypes.c_int32 * n)()
    check(load().fgs_stage_timing_read(ms, cnt), "fgs_stage_timing_read")
    return {STAGES[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}
