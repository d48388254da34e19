"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/*.h declares; plan/layout arithmetic (no GPU compute calls)."""
import ctypes
import glob
import os
import re

import pytest

from helpers import ROOT


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b(fgs_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from fresnel_amd import build
    path = build.build()
    return ctypes.CDLL(path)


def test_library_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert "fgs_forward" in names and "fgs_backward" in names
    for n in names:
        assert hasattr(lib, n), f"libfgs_hip.so does not export {n}"


def test_binding_symbol_list_matches_header():
    from fresnel_amd import _binding
    assert sorted(_binding.EXPORTED_SYMBOLS) == declared_symbols()


def test_workspace_and_layout():
    from fresnel_amd import _binding as B
    d = B.make_dims(2, 1000, 100, 72, tuning=dict(tile_w=16))
    saved, scratch = B.workspace_bytes(d)
    L = B.saved_layout(d)
    assert L.tiles_x == 7 and L.tiles_y == 5 and L.tile_w == 16
    # 64-px radius cap: bbox spans <= 10 tile columns, clipped to the 7x5 tile grid
    assert L.dup_capacity == 2 * 1000 * 7 * 5
    assert L.total_bytes == saved and saved > 0 and scratch > 0
    offs = [L.rec, L.depth_key, L.tile_count, L.order, L.counters, L.ranges, L.dup_ids, L.pix_state]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    # 32 x 16 tiles on request (4 x 5 here; a 64-px radius spans <= 6 tile columns and <= 10 tile rows) ...
    La = B.saved_layout(B.make_dims(2, 1000, 100, 72, tuning=dict(tile_w=32)))
    assert La.tile_w == 32 and La.tiles_x == 4 and La.tiles_y == 5 and La.dup_capacity == 2 * 1000 * 4 * 5
    assert B.saved_layout(B.make_dims(2, 1000, 100, 72)).tile_w == 16
    # ... and by default on the blend path from 512-pixel-wide frames on
    d2 = B.make_dims(8, 32768, 512, 512)
    L2 = B.saved_layout(d2)
    assert L2.tile_w == 32 and L2.dup_capacity == 8 * 32768 * 60 and L2.tiles_x == 16 and L2.tiles_y == 32
    assert B.saved_layout(B.make_dims(8, 32768, 512, 512, tuning=dict(tile_w=16))).dup_capacity == 8 * 32768 * 100
    # the phase path, the row-split forward and saturation_skip keep 16 x 16 tiles and refuse 32
    assert B.saved_layout(B.make_dims(8, 1000, 512, 512, use_phase=True)).tile_w == 16
    assert B.saved_layout(B.make_dims(8, 1000, 512, 512, saturation_skip=True)).tile_w == 16
    with pytest.raises(B.FgsError):
        B.workspace_bytes(B.make_dims(2, 1000, 100, 72, use_phase=True, tuning=dict(tile_w=32)))
    with pytest.raises(B.FgsError):
        B.workspace_bytes(B.make_dims(2, 1000, 100, 72, tuning=dict(tile_w=24)))


def test_invalid_dims_are_rejected():
    from fresnel_amd import _binding as B
    with pytest.raises(B.FgsError):
        B.workspace_bytes(B.make_dims(0, 10, 16, 16))
    with pytest.raises(B.FgsError):
        B.workspace_bytes(B.make_dims(2, 10, 16, 16, num_cameras=3))


def test_renderer_refuses_cpu_tensors():
    """The product path has no CPU fallback: it must fail loudly, not route to the oracle."""
    import torch
    from fresnel_amd import _binding as B
    from fresnel_amd.renderer import Camera, TileBasedRenderer
    r = TileBasedRenderer(32, 32)
    cam = Camera(25.6, 25.6, 16, 16, 32, 32)
    with pytest.raises(B.FgsError):
        r(torch.zeros(4, 3), torch.ones(4, 3), torch.ones(4, 4), torch.ones(4, 3), torch.ones(4), cam)


def test_shipped_library_reads_no_environment(lib):
    """The product library takes every switch through FgsDims: it must not import getenv /
    secure_getenv, so no environment change can put a forward and its backward on different layouts."""
    import subprocess
    from fresnel_amd import build
    out = subprocess.run(["nm", "-D", "--undefined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out, "libfgs_hip.so imports getenv"
    for src in glob.glob(os.path.join(ROOT, "fresnel_amd", "csrc", "*")):
        assert "getenv" not in open(src).read(), src


def test_tuning_fields_are_validated_and_change_only_the_split():
    from fresnel_amd import _binding as B
    base = B.saved_layout(B.make_dims(8, 32768, 512, 512))
    assert base.seg_len == 128
    assert B.saved_layout(B.make_dims(2, 32768, 512, 512)).seg_len == 64
    forced = B.saved_layout(B.make_dims(2, 32768, 512, 512, tuning=dict(seg_len=128)))
    assert forced.seg_len == 128 and forced.dup_capacity == B.saved_layout(B.make_dims(2, 32768, 512, 512)).dup_capacity
    for bad in (dict(seg_len=32), dict(fwd_variant=3), dict(fwd_variant=-8), dict(bin_mode=3)):
        with pytest.raises(B.FgsError):
            B.workspace_bytes(B.make_dims(2, 1000, 64, 64, tuning=bad))
    # the row-split forward (saturation_skip / fwd_variant < 0) stages 128-entry chunks
    with pytest.raises(B.FgsError):
        B.workspace_bytes(B.make_dims(2, 1000, 64, 64, tuning=dict(fwd_variant=-2, seg_len=64)))
    assert B.saved_layout(B.make_dims(2, 1000, 64, 64, saturation_skip=True)).seg_len == 128
    # direct binning needs <= 4096 tiles per image
    with pytest.raises(B.FgsError):
        B.workspace_bytes(B.make_dims(1, 100, 2048, 2048, tuning=dict(bin_mode=1)))
    B.workspace_bytes(B.make_dims(1, 100, 2048, 2048, tuning=dict(bin_mode=2)))
    # ... and <= 512 tile columns + rows (the mask binning keeps one mask line per column / row in LDS): a
    # 8192 x 16 frame (512 + 1 lines) takes the radix path
    with pytest.raises(B.FgsError):
        B.workspace_bytes(B.make_dims(1, 100, 8192, 16, tuning=dict(bin_mode=1, tile_w=16)))
    B.workspace_bytes(B.make_dims(1, 100, 8192, 16, tuning=dict(tile_w=16)))
    B.workspace_bytes(B.make_dims(1, 100, 8176, 16, tuning=dict(bin_mode=1, tile_w=16)))
    B.workspace_bytes(B.make_dims(1, 100, 8192, 16, tuning=dict(bin_mode=1, tile_w=32)))  # 256 + 1 lines of 32 x 16 tiles
    B.workspace_bytes(B.make_dims(8, 100, 8192, 16, tuning=dict(bin_mode=1)))  # 4096 16 x 16 tiles: 32 x 16 automatically


def test_new_entries_validate_arguments_without_a_gpu(lib):
    """The round-2 entry points reject bad shapes / null pointers before touching the device (rc = FGS_EINVAL = -1,
    message in fgs_last_error): gather hand-off, Helmholtz loss, propagator workspace."""
    lib.fgs_last_error.restype = ctypes.c_char_p
    i32, vp, f32 = ctypes.c_int32, ctypes.c_void_p, ctypes.c_float
    lib.fgs_gather_forward.argtypes = [i32] * 4 + [vp] * 14
    lib.fgs_gather_backward.argtypes = [i32] * 4 + [vp] * 14
    nul = [None] * 14
    assert lib.fgs_gather_forward(2, 100, 200, 0, *nul) == -1            # n_out > n_in
    assert b"invalid dims" in lib.fgs_last_error()
    assert lib.fgs_gather_forward(2, 100, 50, 2, *nul) == -1             # phase_channels must be 0 | 1 | 3
    assert lib.fgs_gather_forward(2, 100, 50, 0, *nul) == -1             # null indices
    assert b"null" in lib.fgs_last_error()
    assert lib.fgs_gather_backward(0, 100, 50, 0, *nul) == -1
    lib.fgs_helmholtz_loss_forward.argtypes = [i32, i32, i32, f32, f32] + [vp] * 5
    assert lib.fgs_helmholtz_loss_forward(1, 8, 8, 0.05, 0.0, *([None] * 5)) == -1
    lib.fgs_asm_propagate_workspace_bytes.argtypes = [i32, i32, i32, ctypes.POINTER(ctypes.c_size_t)]
    assert lib.fgs_asm_propagate_workspace_bytes(0, 8, 1, None) == -1
    lib.fgs_reduction_scratch_bytes.restype = ctypes.c_size_t
    assert lib.fgs_reduction_scratch_bytes() >= 8192


def test_product_library_is_not_an_experiment_build(lib):
    """fgs_version() names experiment builds (python -m fresnel_amd.build --define ...: FGS_EXPERIMENT_BUILD + the list of defines,
    fgs_api.hip): the library the tests, the bench and the sweeps load must be the product.  The timing-only switches of the
    compositing unit refuse to compile without that marker, and the wrong-results switch of round 4
    (FGS_WHATIF_HALF_REDUCTIONS) is gone from the source."""
    lib.fgs_version.restype = ctypes.c_char_p
    v = lib.fgs_version().decode()
    assert v.startswith("fgs-hip") and "EXPERIMENT" not in v, v
    src = open(os.path.join(ROOT, "fresnel_amd", "csrc", "fgs_composite.hip")).read()
    assert "FGS_WHATIF" not in src and "#error" in src and "FGS_EXPERIMENT_BUILD" in src
    import fresnel_amd.build as fb
    assert "FGS_EXPERIMENT_BUILD" in open(fb.__file__).read()
