"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and against the
golden vectors generated from the reference renderer (DR:412-686).

Bar (BASELINE.json north_star / SURVEY §8c):
  * integer stages -- visibility, fp64 bbox, canonical depth order, per-tile lists: BIT-EXACT;
  * rendered RGB / depth and all gradients: max|d| <= 1e-4 * max|ref| per tensor.
"""
import numpy as np
import pytest
import torch

from helpers import TBR_CASES, load_golden, oracle_camera, rel_to_max, synth_aniso, synth_saag

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _cuda():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch.device("cuda:0")


def _camera_from_golden(g):
    from fresnel_amd.renderer import Camera
    W, H = [int(v) for v in g["size"]]
    fx, fy, cx, cy, near, far = [float(v) for v in g["intr"]]
    cam = Camera(fx, fy, cx, cy, W, H, near, far)
    cam.set_view(torch.from_numpy(g["view"].astype(np.float32)))
    return cam


def _hip_render(arrs, cam, W, H, bg, phases=None, use_phase=False, amp=0.25, grads=None, tuning=None):
    """arrs: list of numpy (N,.) or (B,N,.) arrays.  Returns dict of numpy results.
    `tuning`: FgsDims work-split overrides (seg_len / fwd_variant / bin_mode)."""
    from fresnel_amd.renderer import TileBasedRenderer
    dev = _cuda()
    ts = [torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(grads is not None) for a in arrs]
    ph = None
    if phases is not None:
        ph = torch.from_numpy(phases).to(dev).requires_grad_(grads is not None)
    ren = TileBasedRenderer(W, H, background=tuple(float(b) for b in bg), use_phase_blending=use_phase,
                            phase_amplitude=amp)
    ren.tuning = tuning
    img, dep = ren(*ts, cam, return_depth=True, phases=ph)
    out = dict(image=img.detach().cpu().numpy(), depth=dep.detach().cpu().numpy())
    if grads is not None:
        gI, gD = grads
        loss = (img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()
        loss.backward()
        for n, t in zip(["positions", "scales", "rotations", "colors", "opacities"], ts):
            out["grad_" + n] = t.grad.detach().cpu().numpy()
        if ph is not None:
            out["grad_phases"] = ph.grad.detach().cpu().numpy()
    return out


def _hip_stages(arrs, cam, W, H, bg=(0, 0, 0), tuning=None):
    """Integer stages of one forward (B,N,.) via the raw C-ABI entry; numpy views."""
    from fresnel_amd import renderer as R
    dev = _cuda()
    ts = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]
    cfg = R._Cfg(W, H, bg, 64, False, 0.25, tuning=tuning)
    camt = R.pack_cameras(cam, dev)
    img, dep, saved, dims, _ = R.forward_raw(*ts, None, camt, cfg)
    torch.cuda.synchronize()
    st = R.inspect_saved(saved, dims)
    out = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in st.items()}
    out["image"], out["depth"] = img.cpu().numpy(), dep.cpu().numpy()
    return out


def _oracle(arrs, ocam, bg, phases=None, amp=0.25):
    from oracle import fgs_oracle as orc
    return orc.render(*arrs, ocam, bg=bg, phases=phases, phase_amp=amp)


def _check_integer_stages(st, b, r, W, H):
    """HIP integer stages of image b vs oracle Rendered r: all bit-exact."""
    from oracle import fgs_oracle as orc
    N = r.pos.shape[0]
    rec = st["rec"][b]
    key = st["depth_key"][b].view(np.uint32)
    vis_h = (key != 0xFFFFFFFF)
    assert np.array_equal(vis_h, r.proj["visible"].astype(bool)), "visibility differs"
    bbx = np.ascontiguousarray(rec[:, 10]).view(np.uint32)
    bby = np.ascontiguousarray(rec[:, 11]).view(np.uint32)
    bbox_h = np.stack([bbx & 0xFFFF, bbx >> 16, bby & 0xFFFF, bby >> 16], 1).astype(np.int32)
    assert np.array_equal(bbox_h[vis_h], r.proj["bbox"][vis_h]), "bbox differs"
    # canonical depth order: visible subsequence of the HIP order == oracle's
    order_h = st["order"][b]
    nv = int(vis_h.sum())
    assert np.array_equal(order_h[:nv], r.vis_sorted), "depth order differs"
    # the projected floats feeding those decisions are bit-identical too (canonical fp32)
    assert np.array_equal(np.ascontiguousarray(rec[vis_h, 0:2]).view(np.uint32), np.ascontiguousarray(r.proj["mean2d"][vis_h]).view(np.uint32))
    assert np.array_equal(np.ascontiguousarray(rec[vis_h, 9]).view(np.uint32), np.ascontiguousarray(r.proj["depth"][vis_h]).view(np.uint32))
    # per-tile lists
    ranges_o, ids_o = orc.tile_lists(r.vis_sorted, r.proj["bbox"], W, H, 16, tile_w=int(st["layout"].tile_w))
    T = len(ranges_o) - 1
    rg = st["ranges"][b]
    dup = st["dup_ids"]
    for t in range(T):
        s, e = int(rg[t, 0]), int(rg[t, 1])
        exp = ids_o[ranges_o[t]:ranges_o[t + 1]]
        assert e - s == len(exp), f"tile {t}: list length {e - s} != {len(exp)}"
        if len(exp):
            assert np.array_equal(dup[s:e] - b * N, exp), f"tile {t}: list differs"


@pytest.mark.parametrize("tile_w", [16, 32])
@pytest.mark.parametrize("case", [c for c in TBR_CASES if not c.startswith("G6")])
def test_golden_forward_backward(case, tile_w):
    g = load_golden(case)
    W, H = [int(v) for v in g["size"]]
    arrs = [g[k] for k in ["positions", "scales", "rotations", "colors", "opacities"]]
    out = _hip_render(arrs, _camera_from_golden(g), W, H, g["background"], grads=(g["gI"], g["gD"]),
                      tuning=dict(tile_w=tile_w))
    assert np.abs(out["image"] - g["image"]).max() <= TOL * max(1.0, float(np.abs(g["image"]).max()))
    assert rel_to_max(out["depth"], g["depth"]) <= TOL
    for k in ["positions", "scales", "rotations", "colors", "opacities"]:
        assert rel_to_max(out["grad_" + k], g["grad_" + k]) <= TOL, k


@pytest.mark.parametrize("tile_w", [16, 32])
@pytest.mark.parametrize("case", [c for c in TBR_CASES if not c.startswith("G6")])
def test_golden_integer_stages(case, tile_w):
    g = load_golden(case)
    W, H = [int(v) for v in g["size"]]
    arrs = [g[k] for k in ["positions", "scales", "rotations", "colors", "opacities"]]
    st = _hip_stages([a[None] for a in arrs], _camera_from_golden(g), W, H, g["background"], tuning=dict(tile_w=tile_w))
    assert int(st["layout"].tile_w) == tile_w
    r = _oracle(arrs, oracle_camera(g), g["background"])
    _check_integer_stages(st, 0, r, W, H)
    # and against the reference's own integer stages stored in the fixture
    vis = g["visible"].astype(bool)
    key = st["depth_key"][0].view(np.uint32)
    assert np.array_equal(key != 0xFFFFFFFF, vis)
    ref_order = g["depth_order"][vis[g["depth_order"]]]
    assert np.array_equal(st["order"][0][:int(vis.sum())], ref_order)


def test_g3_all_behind_camera_gives_background_and_zero_grads():
    g = load_golden("G3_behind64_64")
    W, H = [int(v) for v in g["size"]]
    arrs = [g[k] for k in ["positions", "scales", "rotations", "colors", "opacities"]]
    out = _hip_render(arrs, _camera_from_golden(g), W, H, g["background"], grads=(g["gI"], g["gD"]))
    for ch in range(3):
        assert np.all(out["image"][ch] == g["background"][ch])
    assert not out["depth"].any()
    for k in ["positions", "scales", "rotations", "colors", "opacities"]:
        assert not out["grad_" + k].any()


@pytest.mark.parametrize("N,R,seed", [(2048, 128, 11), (8192, 256, 12)])
def test_synthetic_saag_vs_oracle(N, R, seed):
    """create_dummy_saag distribution (TGD:760-778); N=8192 @256x256 is BASELINE config 2's
    per-image shape."""
    from oracle import fgs_oracle as orc
    arrs = list(synth_saag(N, seed))
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * R, 0.8 * R, R / 2, R / 2, R, R)
    from fresnel_amd.renderer import Camera
    cam = Camera(0.8 * R, 0.8 * R, R / 2, R / 2, R, R)
    r = _oracle(arrs, ocam, (0.0, 0.0, 0.0))
    st = _hip_stages([a[None] for a in arrs], cam, R, R)
    _check_integer_stages(st, 0, r, R, R)
    rs = np.random.RandomState(seed)
    gI = rs.standard_normal((3, R, R)).astype(np.float32)
    gD = (rs.standard_normal((R, R)) * 0.1).astype(np.float32)
    out = _hip_render(arrs, cam, R, R, (0.0, 0.0, 0.0), grads=(gI, gD))
    assert rel_to_max(out["image"], r.image) <= TOL
    assert rel_to_max(out["depth"], r.depth) <= TOL
    go = orc.render_backward(r, gI, gD)
    for k in ["positions", "scales", "rotations", "colors", "opacities"]:
        assert rel_to_max(out["grad_" + k], go[k]) <= TOL, k


def test_batched_ragged_images_vs_oracle():
    """B=3 images with different content, non-multiple-of-16 frame (partial tiles), per-image
    orbit cameras, clamp-active opacities, bg != 0."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    W, H, N, Bn = 100, 72, 700, 3
    bg = (0.1, 0.2, 0.3)
    per = [synth_aniso(N, 20 + b, opacity_max=1.3) for b in range(Bn)]
    arrs = [np.stack([p[i] for p in per]) for i in range(5)]
    views = []
    for b in range(Bn):
        V = np.eye(4, dtype=np.float32)
        a = 0.15 * (b - 1)
        V[0, 0], V[0, 2], V[2, 0], V[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
        V[0, 3] = 0.1 * b
        views.append(V)
    cams, ocams = [], []
    for V in views:
        c = Camera(80.0, 75.0, W / 2, H / 2, W, H)
        c.set_view(torch.from_numpy(V))
        cams.append(c)
        ocams.append(orc.make_camera(V, 80.0, 75.0, W / 2, H / 2, W, H))
    st = _hip_stages(arrs, cams, W, H, bg)
    rs = np.random.RandomState(5)
    gI = rs.standard_normal((Bn, 3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((Bn, H, W)) * 0.1).astype(np.float32)
    out = _hip_render(arrs, cams, W, H, bg, grads=(gI, gD))
    for b in range(Bn):
        r = _oracle([a[b] for a in arrs], ocams[b], bg)
        _check_integer_stages(st, b, r, W, H)
        assert rel_to_max(out["image"][b], r.image) <= TOL
        assert rel_to_max(out["depth"][b], r.depth) <= TOL
        go = orc.render_backward(r, gI[b], gD[b])
        for k in ["positions", "scales", "rotations", "colors", "opacities"]:
            assert rel_to_max(out["grad_" + k][b], go[k]) <= TOL, (b, k)


def test_single_image_call_matches_reference_signature():
    """(N,.) inputs, positional camera, return_depth False/True -> (3,H,W) / tuple (DR:684-686)."""
    from fresnel_amd.renderer import Camera, TileBasedRenderer
    dev = _cuda()
    pos, scale, quat, col, opa = [torch.from_numpy(a).to(dev) for a in synth_saag(128, 3)]
    cam = Camera(0.8 * 64, 0.8 * 64, 32, 32, 64, 64)
    ren = TileBasedRenderer(64, 64).to(dev)
    img = ren(pos, scale, quat, col, opa, cam)
    assert img.shape == (3, 64, 64) and img.dtype == torch.float32
    img2, dep = ren(pos, scale, quat, col, opa, cam, return_depth=True)
    assert dep.shape == (64, 64) and torch.equal(img, img2)


def test_full_size_properties_config3_shape():
    """BASELINE config-3 per-GPU shape (N=32768 @512x512) through size-independent properties:
    determinism of every integer stage, depth-sorted tile lists, pair count = sum of tile
    overlaps, colour linearity, input-permutation invariance of the image."""
    from fresnel_amd import renderer as R
    from fresnel_amd.renderer import Camera
    dev = _cuda()
    N, S = 32768, 512
    arrs = list(synth_saag(N, 77))
    # distinct depths (needed by the permutation property; spacing >> 1 ulp)
    arrs[0][:, 2] = (-1.2 - 1.6 * np.random.RandomState(2).permutation(N) / N).astype(np.float32)
    cam = Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
    st1 = _hip_stages([a[None] for a in arrs], cam, S, S)
    st2 = _hip_stages([a[None] for a in arrs], cam, S, S)
    D = int(st1["counters"][0])
    assert D == int(st2["counters"][0]) and st1["counters"][1] == 0
    assert np.array_equal(st1["dup_ids"][:D], st2["dup_ids"][:D])
    assert np.array_equal(st1["ranges"], st2["ranges"])
    assert np.array_equal(st1["order"], st2["order"])
    assert np.array_equal(st1["image"], st2["image"])  # forward has no atomics: bitwise repeatable
    # tile lists are depth sorted: rank of consecutive entries increases
    rank = np.empty(N, np.int64)
    rank[st1["order"][0]] = np.arange(N)
    rg = st1["ranges"][0]
    ids = st1["dup_ids"][:D]
    rk = rank[ids]
    starts = np.zeros(D, bool)
    starts[rg[rg[:, 1] > rg[:, 0], 0]] = True
    assert np.all((np.diff(rk) > 0) | starts[1:])
    assert int((rg[:, 1] - rg[:, 0]).sum()) == D
    # duplicates == sum of per-Gaussian tile counts
    assert int(st1["tile_count"].sum()) == D
    # depth segments (work units of the backward): ceil(len / FGS_SEG) units per tile, in tile order
    SEG = int(st1["layout"].seg_len)
    assert SEG in (64, 128)
    nseg = (rg[:, 1] - rg[:, 0] + SEG - 1) // SEG
    U = int(st1["counters"][2])
    assert U == int(nseg.sum()) and U == int(st1["seg_off"][-1])
    assert np.array_equal(st1["seg_off"][:-1], np.concatenate([[0], np.cumsum(nseg)[:-1]]))
    assert np.array_equal(st1["seg_tile"][:U], np.repeat(np.arange(rg.shape[0]), nseg))
    # colour linearity: image(c1 + c2) == image(c1) + image(c2) with black background (pre-clamp safe: colours*0.4)
    c1 = (arrs[3] * 0.4).astype(np.float32)
    c2 = (np.roll(arrs[3], 1, axis=0) * 0.4).astype(np.float32)
    im = lambda c: _hip_stages([arrs[0][None], arrs[1][None], arrs[2][None], c[None], arrs[4][None]], cam, S, S)["image"]
    lhs, rhs = im(c1 + c2), im(c1) + im(c2)
    assert np.abs(lhs - rhs).max() <= 1e-5
    # permutation invariance (depths are distinct): shuffle the Gaussians
    perm = np.random.RandomState(1).permutation(N)
    stp = _hip_stages([a[perm][None] for a in arrs], cam, S, S)
    assert np.abs(stp["image"] - st1["image"]).max() <= 1e-5
    assert np.abs(stp["depth"] - st1["depth"]).max() <= 1e-4 * np.abs(st1["depth"]).max()


def test_direct_binning_equals_radix_binning():
    """Two list builders: the mask binning straight from the bboxes (default for a single layer and <= 4096 tiles
    per image) and the emit + stable radix sort path (kept for the layered ASM keys and larger frames), selected
    here with FgsDims.bin_mode.  Both must produce bit-identical lists, ranges, segment tables and images."""
    from fresnel_amd.renderer import Camera
    N, W, H = 4000, 200, 136
    arrs = [np.stack([a, b]) for a, b in zip(synth_aniso(N, 61, smax=0.2), synth_aniso(N, 62, smax=0.05))]
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    outs = [_hip_stages(arrs, cam, W, H, tuning=dict(bin_mode=m)) for m in (1, 2)]
    D = int(outs[0]["counters"][0])
    assert D == int(outs[1]["counters"][0]) and D > 20000
    assert np.array_equal(outs[0]["dup_ids"][:D], outs[1]["dup_ids"][:D])
    for k in ("ranges", "seg_off", "image", "depth"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("shape", [(3, 200, 136, 1), (8, 256, 256, 1), (5, 72, 40, 2), (1, 64, 64, 1)])
def test_forward_launch_order_is_grouped_per_xcd_and_heavy_first(shape):
    """The forward's launch order (saved.tile_order, scheduling only): a permutation of all (image, tile) lists; its eight
    contiguous ranges -- the ranges fgs_xcd_remap deals to the XCDs -- hold exactly the tiles of eight contiguous index
    ranges (one image per XCD at 8 images, bands of tile rows for fewer), each range longest lists first (quarter-octave
    length buckets).  Round 3: this is what took the forward's FETCH_SIZE from 330 to 78 MB per launch at config 3."""
    from fresnel_amd.renderer import Camera
    Bn, W, H, mode = shape
    arrs = [np.stack(x) for x in zip(*[synth_aniso(1500, 70 + b, smax=0.2) for b in range(Bn)])]
    st = _hip_stages(arrs, Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H), W, H, tuning=dict(bin_mode=mode))
    T = st["ranges"].shape[1]
    n = Bn * T
    order = st["tile_order"].astype(np.int64)
    assert np.array_equal(np.sort(order), np.arange(n))
    lens = (st["ranges"][..., 1] - st["ranges"][..., 0]).reshape(-1).astype(np.int64)
    q, r = n >> 3, n & 7
    start = 0
    for g in range(8):
        cnt = q + 1 if g < r else q
        part = order[start:start + cnt]
        assert np.array_equal(np.sort(part), np.arange(start, start + cnt)), f"group {g} holds foreign tiles"
        L = lens[part]

        def bucket(v):  # quarter-octave of the length, as fgs_bin.hip length_bucket orders them
            if v == 0:
                return -1
            lg = int(v).bit_length() - 1
            frac = (v >> (lg - 2)) & 3 if lg >= 2 else (v << (2 - lg)) & 3
            return lg * 4 + frac
        bk = np.array([bucket(int(v)) for v in L])
        assert np.all(np.diff(bk) <= 0), f"group {g} is not heavy-first"
        start += cnt


@pytest.mark.parametrize("N", [1, 63, 65, 257, 777])
def test_mask_binning_ragged_counts_vs_oracle(N):
    """The mask binning packs the depth ranks into 64-bit words, four words per 256-rank block, lines padded to eight
    words: Gaussian counts that are not multiples of 64 / 256, on a frame that is not a whole number of tiles
    (72 x 40 -> 5 x 3 tiles).  Lists, ranges and order bit-exact against the oracle, two images per call."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    W, H = 72, 40
    a0, a1 = synth_aniso(N, 400 + N, smax=0.25), synth_aniso(N, 900 + N, smax=0.1)
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    st = _hip_stages([np.stack([x, y]) for x, y in zip(a0, a1)], cam, W, H, tuning=dict(bin_mode=1))
    for b, arrs in enumerate((a0, a1)):
        r = _oracle(list(arrs), ocam, (0, 0, 0))
        _check_integer_stages(st, b, r, W, H)
        assert rel_to_max(st["image"][b], r.image) <= TOL


def test_mask_binning_dense_tiles_flush_in_windows():
    """6000 wide Gaussians over 3 x 2 tiles: every tile collects several thousand entries out of ONE chunk of depth
    ranks, more than the 2048 a wave of k_mask_emit parks in LDS per flush -- the windowed flush must keep the list
    order.  Bit-exact against the oracle and against the radix path."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    N, W, H = 6000, 48, 32
    rs = np.random.RandomState(77)
    pos = (rs.randn(N, 3) * [0.3, 0.25, 0.3] + [0, 0, -2.0]).astype(np.float32)
    scale = (0.25 * rs.uniform(0.5, 1.5, (N, 3))).astype(np.float32)
    quat = rs.randn(N, 4).astype(np.float32)
    col = rs.rand(N, 3).astype(np.float32)
    opa = rs.uniform(0.002, 0.02, N).astype(np.float32)
    arrs = [pos, scale, quat, col, opa]
    cam = Camera(0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    outs = [_hip_stages([a[None] for a in arrs], cam, W, H, tuning=dict(bin_mode=m)) for m in (1, 2)]
    lens = outs[0]["ranges"][0][:, 1] - outs[0]["ranges"][0][:, 0]
    assert lens.max() > 2 * 2048
    r = _oracle(arrs, ocam, (0, 0, 0))
    _check_integer_stages(outs[0], 0, r, W, H)
    D = int(outs[0]["counters"][0])
    assert np.array_equal(outs[0]["dup_ids"][:D], outs[1]["dup_ids"][:D])
    assert np.array_equal(outs[0]["ranges"], outs[1]["ranges"])
    assert np.array_equal(outs[0]["dup_off"], outs[1]["dup_off"])


@pytest.mark.parametrize("N,W,H,scale,min_len", [(20000, 48, 32, 0.25, 2 * 8192), (40000, 200, 136, 0.03, 64)])
def test_mask_binning_block_per_list_vs_oracle(N, W, H, scale, min_len):
    """k_mask_emit_block (launches of <= 8192 lists over more than 8192 Gaussians per image: one block per list, 256 lanes
    over the rank words).  (a) 20 000 wide Gaussians over 3 x 2 tiles: every list collects more than the 8192 entries a
    block parks per flush, so the windowed flush runs; (b) 40 000 small ones: two chunks of 32 768 depth ranks per list.
    Bit-exact against the oracle and against the radix path, two images per call."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    rs = np.random.RandomState(N)
    batch = []
    for b in range(2):
        pos = (rs.randn(N, 3) * [0.3, 0.25, 0.3] + [0, 0, -2.0]).astype(np.float32)
        sc = (scale * rs.uniform(0.5, 1.5, (N, 3))).astype(np.float32)
        batch.append([pos, sc, rs.randn(N, 4).astype(np.float32), rs.rand(N, 3).astype(np.float32),
                      rs.uniform(0.002, 0.02, N).astype(np.float32)])
    arrs = [np.stack([batch[0][i], batch[1][i]]) for i in range(5)]
    cam = Camera(0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    outs = [_hip_stages(arrs, cam, W, H, tuning=dict(bin_mode=m)) for m in (1, 2)]
    lens = outs[0]["ranges"][:, :, 1] - outs[0]["ranges"][:, :, 0]
    assert lens.max() > min_len
    for b in range(2):
        _check_integer_stages(outs[0], b, _oracle(batch[b], ocam, (0, 0, 0)), W, H)
    D = int(outs[0]["counters"][0])
    assert np.array_equal(outs[0]["dup_ids"][:D], outs[1]["dup_ids"][:D])
    lens2 = outs[1]["ranges"][:, :, 1] - outs[1]["ranges"][:, :, 0]
    assert np.array_equal(lens, lens2)
    assert np.array_equal(outs[0]["ranges"][lens > 0], outs[1]["ranges"][lens > 0])  # (an empty list's start is free)
    assert np.array_equal(outs[0]["dup_off"], outs[1]["dup_off"])


def test_count_pairs_equals_the_bbox_areas():
    """fgs_count_pairs (the benchmark's unit of work, SURVEY 8d: composited Gaussian-pixels = sum of the visible
    Gaussians' integer bbox areas) against numpy on the saved records and against the oracle's P."""
    import ctypes
    from fresnel_amd import _binding as B, renderer as R
    from fresnel_amd.renderer import Camera
    from oracle import fgs_oracle as orc
    dev = _cuda()
    N, W, H = 5000, 200, 136
    a0, a1 = synth_aniso(N, 91, smax=0.2), synth_aniso(N, 92, smax=0.05)
    ts = [torch.from_numpy(np.stack([x, y])).to(dev) for x, y in zip(a0, a1)]
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    cfg = R._Cfg(W, H, (0, 0, 0), 64, False, 0.25)
    _, _, saved, dims, _ = R.forward_raw(*ts, None, R.pack_cameras(cam, dev), cfg)
    pairs = torch.zeros(1, dtype=torch.int64, device=dev)
    B.check(B.load().fgs_count_pairs(ctypes.byref(dims), ctypes.c_void_p(saved.data_ptr()), ctypes.c_void_p(pairs.data_ptr()),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "fgs_count_pairs")
    st = R.inspect_saved(saved, dims)
    rec = st["rec"].cpu().numpy()
    cnt = st["tile_count"].cpu().numpy()
    bbx = np.ascontiguousarray(rec[..., 10]).view(np.uint32).astype(np.int64)
    bby = np.ascontiguousarray(rec[..., 11]).view(np.uint32).astype(np.int64)
    area = ((bbx >> 16) - (bbx & 0xFFFF)) * ((bby >> 16) - (bby & 0xFFFF))
    expect = int(area[cnt != 0].sum())
    assert int(pairs.item()) == expect and expect > 100000
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    assert expect == sum(int(orc.render(*a, ocam, keep_pairs=False).P) for a in (a0, a1))


@pytest.mark.parametrize("W,H,N,smax,fwd,seg", [(72, 40, 900, 0.15, 4, 64), (200, 136, 4000, 0.2, 2, 128), (48, 32, 1500, 0.5, 1, 64),
                                                (33, 17, 300, 0.3, 4, 128), (48, 32, 1500, 0.5, 8, 64)])
def test_wide_tiles_vs_oracle(W, H, N, smax, fwd, seg):
    """32 x 16 tiles (FgsDims.tile_w = 32: eight sub-tiles per lane in the backward, two waves per list part -- one per
    16 x 16 half -- in the forward, [5][8][64] checkpoint slots) on frames that are not whole numbers of tiles, with every
    forward split and both segment lengths, two images per call, some opacities above the clamp and below zero: integer
    stages bit-exact, image / depth / all gradients against the oracle, and bitwise the same lists' CONTENT as 16 x 16
    tiles would give after merging (checked through the oracle's rectangular tile lists)."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    a0 = list(synth_aniso(N, 500 + N, smax=smax, opacity_max=1.1))
    a1 = list(synth_aniso(N, 900 + N, smax=smax / 3))
    a0[4][::7] = -0.2
    a0[4][3::11] = 0.995
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    bg = (0.3, 0.1, 0.2)
    tuning = dict(tile_w=32, fwd_variant=fwd, seg_len=seg)
    batch = [np.stack([x, y]) for x, y in zip(a0, a1)]
    st = _hip_stages(batch, cam, W, H, bg, tuning=tuning)
    assert int(st["layout"].tile_w) == 32 and int(st["layout"].tiles_x) == (W + 31) // 32
    rs = np.random.RandomState(N)
    gI = rs.standard_normal((2, 3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((2, H, W)) * 0.1).astype(np.float32)
    out = _hip_render(batch, cam, W, H, bg, grads=(gI, gD), tuning=tuning)
    for b, arrs in enumerate((a0, a1)):
        r = _oracle(arrs, ocam, bg)
        _check_integer_stages(st, b, r, W, H)
        go = orc.render_backward(r, gI[b], gD[b])
        assert rel_to_max(out["image"][b], r.image) <= TOL and rel_to_max(out["depth"][b], r.depth) <= TOL
        for k in ["positions", "scales", "rotations", "colors", "opacities"]:
            assert rel_to_max(out["grad_" + k][b], go[k]) <= TOL, (b, k)


def test_automatic_tile_width_vs_oracle():
    """FgsDims.tile_w = 0: 32 x 16 tiles from 512-pixel-wide frames on when the call has >= 3072 16 x 16 tiles, on the
    blend path with the depth-split forward only; 16 x 16 otherwise, on the phase path and with saturation_skip.  Sixteen
    520 x 120 frames (a ragged last wide tile) against the oracle with nothing forced."""
    from oracle import fgs_oracle as orc
    from fresnel_amd import _binding as B
    from fresnel_amd.renderer import Camera
    assert B.saved_layout(B.make_dims(8, 1000, 512, 512)).tile_w == 32
    assert B.saved_layout(B.make_dims(4, 1000, 512, 512)).tile_w == 32
    assert B.saved_layout(B.make_dims(3, 1000, 512, 512)).tile_w == 32
    assert B.saved_layout(B.make_dims(2, 1000, 512, 512)).tile_w == 16
    assert B.saved_layout(B.make_dims(1, 1000, 1024, 1024)).tile_w == 32
    assert B.saved_layout(B.make_dims(64, 1000, 511, 512)).tile_w == 16
    assert B.saved_layout(B.make_dims(8, 1000, 512, 512, use_phase=True)).tile_w == 16
    assert B.saved_layout(B.make_dims(8, 1000, 512, 512, saturation_skip=True)).tile_w == 16
    assert B.saved_layout(B.make_dims(8, 1000, 512, 512, tuning=dict(fwd_variant=-2))).tile_w == 16
    with pytest.raises(B.FgsError):
        B.saved_layout(B.make_dims(8, 1000, 512, 512, use_phase=True, tuning=dict(tile_w=32)))
    W, H, N, Bn = 520, 120, 400, 16
    per_image = [list(synth_aniso(N, 770 + b, smax=0.1, spread=0.9)) for b in range(Bn)]
    batch = [np.stack([a[i] for a in per_image]) for i in range(5)]
    cam = Camera(0.5 * W, 0.5 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.5 * W, 0.5 * W, W / 2, H / 2, W, H)
    bg = (0.2, 0.4, 0.1)
    st = _hip_stages(batch, cam, W, H, bg)
    assert int(st["layout"].tile_w) == 32 and int(st["layout"].tiles_x) == 17
    rs = np.random.RandomState(5)
    gI = rs.standard_normal((Bn, 3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((Bn, H, W)) * 0.1).astype(np.float32)
    out = _hip_render(batch, cam, W, H, bg, grads=(gI, gD))
    for b in (0, 7, 15):
        r = _oracle(per_image[b], ocam, bg)
        _check_integer_stages(st, b, r, W, H)
        go = orc.render_backward(r, gI[b], gD[b])
        assert rel_to_max(out["image"][b], r.image) <= TOL and rel_to_max(out["depth"][b], r.depth) <= TOL
        for k in ["positions", "scales", "rotations", "colors", "opacities"]:
            assert rel_to_max(out["grad_" + k][b], go[k]) <= TOL, (b, k)


def test_wide_frame_direct_binning_division_is_exact():
    """A frame of 80 tile columns with Gaussians whose bbox spans > 64 of them (radius cap 700): the tile-row
    division of the direct binning's scatter must be exact for any width (the rounded-up reciprocal alone
    is exact only for <= 64 columns).  Direct lists == radix lists == oracle lists."""
    from oracle import fgs_oracle as orc
    from fresnel_amd import renderer as R
    from fresnel_amd.renderer import Camera
    dev = _cuda()
    W, H, N = 1280, 800, 40
    rs = np.random.RandomState(12)
    pos = (rs.standard_normal((N, 3)) * [0.4, 0.25, 0.1] + [0, 0, -2.0]).astype(np.float32)
    scale = (rs.uniform(0.25, 0.6, (N, 3))).astype(np.float32)
    quat = rs.standard_normal((N, 4)).astype(np.float32)
    col = rs.random_sample((N, 3)).astype(np.float32)
    opa = rs.uniform(0.05, 0.3, N).astype(np.float32)
    arrs = [pos, scale, quat, col, opa]
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    r = orc.render(*arrs, ocam, max_radius=700.0, keep_pairs=False)
    widths = (r.proj["bbox"][:, 1] - 1) // 16 - r.proj["bbox"][:, 0] // 16 + 1
    assert widths[r.proj["visible"].astype(bool)].max() > 64
    ts = [torch.from_numpy(a[None]).to(dev) for a in arrs]
    camt = R.pack_cameras(cam, dev)
    outs = []
    for mode in (1, 2):
        cfg = R._Cfg(W, H, (0, 0, 0), 700, False, 0.25, tuning=dict(bin_mode=mode))
        img, dep, saved, dims, _ = R.forward_raw(*ts, None, camt, cfg)
        torch.cuda.synchronize()
        st = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in R.inspect_saved(saved, dims).items()}
        st["image"] = img.cpu().numpy()
        outs.append(st)
    _check_integer_stages(outs[0], 0, r, W, H)
    D = int(outs[0]["counters"][0])
    assert np.array_equal(outs[0]["dup_ids"][:D], outs[1]["dup_ids"][:D])
    assert np.array_equal(outs[0]["ranges"], outs[1]["ranges"])
    assert rel_to_max(outs[0]["image"][0], r.image) <= TOL


def test_negative_and_saturating_opacities_vs_oracle():
    """alpha = clamp(G * opacity, 0, 0.99) (DR:646): a negative opacity contributes nothing and gets zero
    gradients (the kernels drop such records at staging time), opacities above 0.98 take the clamped path,
    opacity exactly 0 is a live record with a non-zero opacity gradient."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    N, W, H = 600, 80, 64
    arrs = list(synth_aniso(N, 33, opacity_max=1.2))
    arrs[4][::5] = -np.abs(arrs[4][::5]) - 0.05  # negative
    arrs[4][1::7] = 0.0                           # exactly zero
    arrs[4][2::11] = 0.985                        # between the no-clamp threshold (0.98) and the clamp (0.99)
    cam = Camera(0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    bg = (0.4, 0.1, 0.3)
    r = _oracle(arrs, ocam, bg)
    rs = np.random.RandomState(34)
    gI = rs.standard_normal((3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    out = _hip_render(arrs, cam, W, H, bg, grads=(gI, gD))
    go = orc.render_backward(r, gI, gD)
    assert rel_to_max(out["image"], r.image) <= TOL and rel_to_max(out["depth"], r.depth) <= TOL
    for k in ["positions", "scales", "rotations", "colors", "opacities"]:
        assert rel_to_max(out["grad_" + k], go[k]) <= TOL, k
    neg = arrs[4] < 0
    for k in ["positions", "scales", "rotations", "colors", "opacities"]:
        assert not np.any(out["grad_" + k][neg]), k  # exactly zero, as in the reference's clamp backward


def test_backward_is_bitwise_reproducible_at_config3_size():
    """No atomics anywhere on the gradient path (one row per (tile, Gaussian) duplicate, fixed-order sums):
    two runs of forward + backward on BASELINE config-3's per-GPU shape give bit-identical gradients."""
    from fresnel_amd.renderer import Camera
    N, S = 32768, 512
    arrs = list(synth_saag(N, 123))
    cam = Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
    rs = np.random.RandomState(6)
    gI = rs.standard_normal((3, S, S)).astype(np.float32)
    gD = (rs.standard_normal((S, S)) * 0.1).astype(np.float32)
    a = _hip_render(arrs, cam, S, S, (0.0, 0.0, 0.0), grads=(gI, gD))
    b = _hip_render(arrs, cam, S, S, (0.0, 0.0, 0.0), grads=(gI, gD))
    for k in a:
        assert np.array_equal(a[k], b[k]), k
        assert np.isfinite(a[k]).all(), k


def test_long_lists_many_depth_segments_vs_oracle():
    """Tiles whose lists span several depth segments (64 entries each at this size): the backward restarts
    every segment from the forward's checkpoint, so gradients must still match the oracle, which
    walks each list in one piece.  1500 wide Gaussians over a 48x32 frame -> ~1000 entries per tile."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    N, W, H = 1500, 48, 32
    rs = np.random.RandomState(5)
    pos = (rs.randn(N, 3) * [0.25, 0.2, 0.3] + [0, 0, -2.0]).astype(np.float32)
    scale = (0.25 * rs.uniform(0.5, 1.5, (N, 3))).astype(np.float32)
    quat = rs.randn(N, 4).astype(np.float32)
    col = rs.rand(N, 3).astype(np.float32)
    opa = rs.uniform(0.005, 0.05, N).astype(np.float32)  # low opacity: the deep segments still matter
    opa[::97] = 0.995                                     # a few entries where the alpha clamp binds
    arrs = [pos, scale, quat, col, opa]
    cam = Camera(0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    bg = (0.2, 0.5, 0.7)
    st = _hip_stages([a[None] for a in arrs], cam, W, H, bg)
    lens = st["ranges"][0][:, 1] - st["ranges"][0][:, 0]
    assert lens.max() > 4 * 128, "test must cover tiles with several depth segments"
    assert int(st["layout"].seg_len) == 64  # small problem: the shorter segments (128 is covered by
    # test_forward_variants_agree and tests/test_hip_headline.py)
    r = _oracle(arrs, ocam, bg)
    _check_integer_stages(st, 0, r, W, H)
    gI = rs.standard_normal((3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    out = _hip_render(arrs, cam, W, H, bg, grads=(gI, gD))
    assert rel_to_max(out["image"], r.image) <= TOL
    assert rel_to_max(out["depth"], r.depth) <= TOL
    go = orc.render_backward(r, gI, gD)
    for k in ["positions", "scales", "rotations", "colors", "opacities"]:
        assert rel_to_max(out["grad_" + k], go[k]) <= TOL, k


@pytest.mark.parametrize("use_phase", [False, True])
def test_forward_variants_agree(use_phase):
    """The forward picks its work split from the launch size; FgsDims.fwd_variant / seg_len force one.  Blend path:
    the depth-split kernel with 1, 2, 4, 8 or 16 list parts per tile (partial results composed with
    (C,T)o(C',T') = (C + T C', T T'), the backward re-bases part-local checkpoints), with 64- or 128-entry depth
    segments, or the row-split kernel with 1, 2 or 4 waves per tile.  Phase path: ONE work split since round 4 (one wave
    per 8 x 8 sub-tile, wave-private compacted lists, checkpoint groups of eight touched entries) -- the automatic choice
    and its explicit name (fwd_variant 4) are the same kernels, other values are refused; this long-list scene (~80
    entries per list, several scan blocks and checkpoint groups per sub-tile) is its oracle check.  All
    variants must agree on image, depth and -- through the saved state and checkpoints -- on every gradient, to
    1e-5 of max (composition order and FMA contraction differ), and with the ORACLE to the parity tolerance.
    The lists are long enough for several parts: ~3000 entries over 36 tiles."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    N, S = 3000, 96
    arrs = list(synth_aniso(N, 9))
    cam = Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
    rs = np.random.RandomState(4)
    gI = rs.standard_normal((3, S, S)).astype(np.float32)
    gD = (rs.standard_normal((S, S)) * 0.1).astype(np.float32)
    phases = rs.uniform(0, 1, N).astype(np.float32) if use_phase else None
    if use_phase:
        variants = [dict(), dict(fwd_variant=4)]
        from fresnel_amd import _binding as B
        for bad in (1, 2, 8):
            with pytest.raises(B.FgsError):
                B.saved_layout(B.make_dims(1, N, S, S, use_phase=True, tuning=dict(fwd_variant=bad)))
    else:
        variants = ([dict(fwd_variant=p, seg_len=sl) for p in (1, 2, 4) for sl in (64, 128)] +
                    [dict(fwd_variant=-w) for w in (1, 2, 4)] +
                    [dict(fwd_variant=8, seg_len=64), dict(fwd_variant=16, seg_len=64), dict(fwd_variant=16, seg_len=128),
                     dict(fwd_variant=8, seg_len=64, tile_w=32), dict(fwd_variant=16, seg_len=64, tile_w=32)])
    bg = (0.1, 0.2, 0.3)
    outs = [_hip_render(arrs, cam, S, S, bg, phases=phases, use_phase=use_phase, grads=(gI, gD), tuning=t)
            for t in variants]
    for t, o in zip(variants[1:], outs[1:]):
        for k in outs[0]:
            assert rel_to_max(o[k], outs[0][k]) <= 1e-5, (t, k)
    r = _oracle(arrs, ocam, bg, phases=phases)
    go = orc.render_backward(r, gI, gD)
    for t, o in zip(variants, outs):
        assert rel_to_max(o["image"], r.image) <= TOL and rel_to_max(o["depth"], r.depth) <= TOL, t
        for k in ["positions", "scales", "rotations", "colors", "opacities"] + (["phases"] if use_phase else []):
            assert rel_to_max(o["grad_" + k], go[k]) <= TOL, (t, k)


@pytest.mark.parametrize("sort_mode", [1, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
@pytest.mark.parametrize("N", [1000, 5000, 8192, 33000, 70000])
def test_depth_sort_key_compression_gives_the_full_key_order(N, sort_mode):
    """Round 4 (BASELINE config 4, "depth-zone sort keys"): with FgsDims.sort_mode bit 0 set the depth sort keeps only the key bits
    that vary over an image's visible Gaussians and runs as many passes as those need.  Round 5: bits 1-3 select the pass
    implementation -- 0 = automatic (images of <= 8192 Gaussians: ONE launch, one block per image, all passes in LDS -- N = 1000, 5000
    and 8192 here, the last one filling its eight chunks per wave exactly; the two-launch 8-bit passes of rounds 1-4 above), 2 = one
    launch per pass, every block forming the segment's histogram itself, 11-bit digits, up to 65 536 Gaussians (N = 70000 falls back
    to two launches), 4 = the same with 8-bit digits, 6 = 8-bit digits with the blocks' histograms handed off between them (published
    words polled with a bounded wait), 8 = the two-launch passes at every size, 10 = as automatic.  Whatever the depths look like, and in EVERY mode, `order` must be the stable argsort of the
    FULL keys (culled keys = 0xFFFFFFFF last, in index order).  One batch, one image of each kind:
      0  zone-snapped depths (8 values: 3 varying bits, ONE live pass), some Gaussians culled;
      1  ordinary depths (~25 varying bits: all four passes);
      2  every Gaussian behind the camera (all culled: nothing varies, the payload is copied by the forced pass);
      3  one single depth, nothing culled (no varying bit, no culled bit);
      4  two depths that differ in ONE low mantissa bit + culled ones (2 bits: one pass);
      5  depths that differ in the exponent byte and the low byte only (16 varying bits in two separate bytes).
    N = 1000 / 5000 / 33000: segments that are not whole projection blocks (key statistics) or whole sort rounds."""
    from fresnel_amd.renderer import Camera
    S = 128
    rs = np.random.RandomState(N)
    per = []
    for img in range(6):
        pos, scale, quat, col, opa = [a.copy() for a in synth_saag(N, 300 + img)]
        pos[:, :2] *= 0.3
        if img == 0:
            pos[:, 2] = -2.0 - 2.0 * (rs.randint(0, 8, N) + 0.5) / 8.0
            pos[rs.rand(N) < 0.05, 0] = 50.0            # far off-screen: culled
        elif img == 2:
            pos[:, 2] = 3.0
        elif img == 3:
            pos[:, 2] = -2.5
        elif img == 4:
            z = np.full(N, -2.5, np.float32)
            z[rs.rand(N) < 0.5] = np.nextafter(np.float32(-2.5), np.float32(-3.0))
            pos[:, 2] = z
            pos[rs.rand(N) < 0.1, 1] = -60.0
        elif img == 5:
            base = np.array([0.75, 1.5, 3.0, 6.0], np.float32)[rs.randint(0, 4, N)]
            low = rs.randint(0, 256, N).astype(np.uint32)
            pos[:, 2] = -(base.view(np.uint32) | low).view(np.float32)
        per.append((pos, scale * 0.2, quat, col, opa))
    arrs = [np.stack([p[i] for p in per]) for i in range(5)]
    st = _hip_stages(arrs, Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S), S, S, tuning=dict(sort_mode=sort_mode))
    keys = st["depth_key"].view(np.uint32)
    kinds = []
    for b in range(6):
        want = np.argsort(keys[b], kind="stable")
        assert np.array_equal(st["order"][b], want.astype(np.int32)), f"image {b}: order is not the stable argsort of the keys"
        vis = keys[b] != 0xFFFFFFFF
        m = (np.bitwise_or.reduce(keys[b][vis]) ^ np.bitwise_and.reduce(keys[b][vis])) if vis.any() else 0
        kinds.append((bin(int(m)).count("1"), int((~vis).sum())))
    # the scenes are what the docstring says they are: (varying bits, culled) per image
    assert kinds[0][0] == 3 and kinds[0][1] > 0 and kinds[1][0] >= 20 and kinds[2] == (0, N) and kinds[3] == (0, 0)
    assert kinds[4][0] == 1 and kinds[4][1] > 0 and kinds[5][0] >= 9


def test_needle_and_disc_gaussians_keep_the_alpha_clamp():
    """Needles and edge-on discs (scale ratio 300:1) make the regularised 2x2 inverse covariance ill-conditioned;
    where it comes out indefinite in fp32, G = exp(-m/2) exceeds 1 and alpha = min(G op, 0.99) binds even for small
    opacities (DR:647).  The backward's no-clamp fast path must not be taken for such records, or it
    would recompute a different transmittance than the forward used.  Checked through a property that is exact
    whatever the conic: the image is LINEAR in the colours, so the colour gradient times a colour step must equal
    the change of the loss -- this compares the w = alpha T of the backward with the forward's, pixel by pixel.
    (Such Gaussians are outside the 1e-4 parity domain, DESIGN section 2: the conic itself differs by rounding
    amplified 1e4x; the oracle comparison is therefore only held to 2e-3, gradients of geometry to finiteness.)"""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    N, W, H = 400, 96, 80
    rs = np.random.RandomState(21)
    pos, scale, quat, col, opa = synth_aniso(N, 22, opacity_max=0.9)
    scale[::2] = np.stack([rs.uniform(0.2, 0.4, N // 2), rs.uniform(0.0008, 0.0015, N // 2),
                           rs.uniform(0.0008, 0.0015, N // 2)], 1).astype(np.float32)      # needles
    scale[1::4] = np.stack([rs.uniform(0.2, 0.3, N // 4), rs.uniform(0.2, 0.3, N // 4),
                            rs.uniform(0.0008, 0.0015, N // 4)], 1).astype(np.float32)     # discs
    col = (col * 0.5).astype(np.float32)
    arrs = [pos, scale, quat, col, opa]
    cam = Camera(0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
    bg = (0.2, 0.1, 0.4)
    gI = rs.standard_normal((3, H, W)).astype(np.float32)
    gD = np.zeros((H, W), np.float32)
    out = _hip_render(arrs, cam, W, H, bg, grads=(gI, gD))
    step = (rs.uniform(0.0, 0.4, col.shape)).astype(np.float32)  # colours stay in [0, 0.9]: the output clamp is idle
    out2 = _hip_render([pos, scale, quat, col + step, opa], cam, W, H, bg)
    lhs = float((gI.astype(np.float64) * (out2["image"].astype(np.float64) - out["image"])).sum())
    rhs = float((out["grad_colors"].astype(np.float64) * step).sum())
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), float(np.abs(out["grad_colors"] * step).sum())), (lhs, rhs)
    r = _oracle(arrs, ocam, bg)
    go = orc.render_backward(r, gI, gD)
    assert rel_to_max(out["image"], r.image) <= 2e-3 and rel_to_max(out["depth"], r.depth) <= 2e-3
    for k in ["colors", "opacities"]:
        assert rel_to_max(out["grad_" + k], go[k]) <= 2e-3, k
    for k in ["positions", "scales", "rotations"]:
        assert np.isfinite(out["grad_" + k]).all(), k


# ------------------------------------------------------------------------------------------
# Phase-blending path (BASELINE config 4: --use_fresnel_zones 8 --use_phase_blending)
# ------------------------------------------------------------------------------------------
def test_golden_g6_phase_forward_and_backward():
    """Forward vs the REFERENCE (DR:629-667); colour gradient vs the reference's own autograd
    (the only backward it can produce); all gradients incl. dL/dphase vs the
    fixture's out-of-place restatement (which reproduced the reference forward exactly)."""
    g = load_golden("G6_phase256_128")
    W, H = [int(v) for v in g["size"]]
    arrs = [g[k] for k in ["positions", "scales", "rotations", "colors", "opacities"]]
    out = _hip_render(arrs, _camera_from_golden(g), W, H, g["background"], phases=g["phases"],
                      use_phase=True, amp=float(g["phase_amplitude"]), grads=(g["gI"], g["gD"]))
    assert rel_to_max(out["image"], g["image"]) <= TOL
    assert rel_to_max(out["depth"], g["depth"]) <= TOL
    assert rel_to_max(out["grad_colors"], g["ref_grad_colors"]) <= TOL
    for k in ["positions", "scales", "rotations", "colors", "opacities", "phases"]:
        assert rel_to_max(out["grad_" + k], g["restated_grad_" + k]) <= TOL, k


def test_phase_zone_depths_batched_vs_oracle():
    """Config-4 style input: depths snapped to 8 zone centres (massive sort ties -> canonical
    stable order), edge-aware scale factors, scalar phases, amp 0.25; B=2, ragged frame."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    W, H, N, Bn = 144, 112, 1500, 2
    rs = np.random.RandomState(44)
    per = []
    for b in range(Bn):
        pos, scale, quat, col, opa = synth_aniso(N, 50 + b, opacity_max=1.0, smin=0.02, smax=0.09)
        zone = rs.randint(0, 8, N)
        pos[:, 2] = (-2.0 - 2.0 * (zone + 0.5) / 8.0).astype(np.float32)
        scale = (scale * rs.uniform(0.5, 1.0, (N, 1))).astype(np.float32)
        per.append((pos, scale, quat, col, opa))
    arrs = [np.stack([p[i] for p in per]) for i in range(5)]
    phases = rs.random_sample((Bn, N)).astype(np.float32)
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    gI = rs.standard_normal((Bn, 3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((Bn, H, W)) * 0.1).astype(np.float32)
    out = _hip_render(arrs, cam, W, H, (0.05, 0.1, 0.15), phases=phases, use_phase=True, amp=0.25, grads=(gI, gD))
    for b in range(Bn):
        r = _oracle([a[b] for a in arrs], ocam, (0.05, 0.1, 0.15), phases=phases[b], amp=0.25)
        assert len(np.unique(r.proj["depth"][r.proj["visible"].astype(bool)])) <= 8
        assert rel_to_max(out["image"][b], r.image) <= TOL
        assert rel_to_max(out["depth"][b], r.depth) <= TOL
        go = orc.render_backward(r, gI[b], gD[b])
        for k in ["positions", "scales", "rotations", "colors", "opacities"]:
            assert rel_to_max(out["grad_" + k][b], go[k]) <= TOL, (b, k)
        assert rel_to_max(out["grad_phases"][b], go["phases"]) <= TOL


def test_phase_rgb_phases_raise_like_the_reference():
    """(N,3) phases crash the reference's TBR phase path: same error class here."""
    from fresnel_amd.renderer import Camera, TileBasedRenderer
    dev = _cuda()
    pos, scale, quat, col, opa = [torch.from_numpy(a).to(dev) for a in synth_saag(64, 3)]
    ren = TileBasedRenderer(32, 32, use_phase_blending=True)
    with pytest.raises(RuntimeError):
        ren(pos, scale, quat, col, opa, Camera(25.6, 25.6, 16, 16, 32, 32), phases=torch.rand(64, 3, device=dev))
    # phases are ignored unless use_phase_blending is set (DR:629)
    ren0 = TileBasedRenderer(32, 32)
    a = ren0(pos, scale, quat, col, opa, Camera(25.6, 25.6, 16, 16, 32, 32), phases=torch.rand(64, device=dev))
    b = ren0(pos, scale, quat, col, opa, Camera(25.6, 25.6, 16, 16, 32, 32))
    assert torch.equal(a, b)


def test_edge_cases_tiny_frames_single_gaussian_and_empty_input():
    """Frame smaller than one tile, a single Gaussian, N not a multiple of anything, a Gaussian far
    larger than the frame (radius cap + full-frame bbox), and N = 0."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera, TileBasedRenderer
    dev = _cuda()
    rs = np.random.RandomState(3)
    for (W, H, N) in [(8, 8, 1), (5, 9, 7), (17, 33, 3), (40, 24, 131)]:
        pos = (rs.standard_normal((N, 3)) * 0.3).astype(np.float32)
        pos[:, 2] -= 1.5
        scale = (rs.random_sample((N, 3)) * 0.4 + 0.02).astype(np.float32)
        quat = rs.standard_normal((N, 4)).astype(np.float32)
        col = rs.random_sample((N, 3)).astype(np.float32)
        opa = rs.random_sample(N).astype(np.float32)
        cam = Camera(0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
        ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * H, W / 2, H / 2, W, H)
        gI = rs.standard_normal((3, H, W)).astype(np.float32)
        gD = rs.standard_normal((H, W)).astype(np.float32)
        out = _hip_render([pos, scale, quat, col, opa], cam, W, H, (0.3, 0.2, 0.1), grads=(gI, gD))
        r = _oracle([pos, scale, quat, col, opa], ocam, (0.3, 0.2, 0.1))
        assert rel_to_max(out["image"], r.image) <= TOL and rel_to_max(out["depth"], r.depth) <= TOL
        go = orc.render_backward(r, gI, gD)
        for k in ["positions", "scales", "rotations", "colors", "opacities"]:
            assert rel_to_max(out["grad_" + k], go[k]) <= TOL, (W, H, N, k)
    ren = TileBasedRenderer(16, 12, background=(0.5, 0.25, 0.125))
    z = lambda *s: torch.zeros(*s, device=dev)
    img, dep = ren(z(0, 3), z(0, 3), z(0, 4), z(0, 3), z(0), Camera(12.8, 9.6, 8, 6, 16, 12), return_depth=True)
    assert img.shape == (3, 12, 16) and torch.all(img[1] == 0.25) and not dep.any()


@pytest.mark.parametrize("W,H,N,smax,seed", [(115, 15, 900, 0.02, 71), (16, 16, 5000, 0.2, 72), (13, 48, 3, 0.02, 73)])
def test_subpixel_and_crowded_gaussians_vs_oracle(W, H, N, smax, seed):
    """Gaussians much smaller than a pixel / than a tile (scale <= 0.02 -> sigma of a fraction of a pixel) and a
    16x16 frame crowded with 5000 of them: the second moments sum dG dx^2 of such Gaussians are tiny against
    |dG| * (tile size)^2, so any formulation that accumulates moments about a far origin loses dL/dscale to
    cancellation (a raw-moment variant of the backward failed exactly these cases of the randomized sweeps, by up to
    4e-2).  All gradients must stay within 1e-4 of the oracle."""
    from oracle import fgs_oracle as orc
    from fresnel_amd.renderer import Camera
    rs = np.random.RandomState(seed)
    arrs = list(synth_aniso(N, seed, opacity_max=1.0, smax=smax))
    bg = tuple(float(x) for x in rs.rand(3))
    cam = Camera(0.9 * W, 0.9 * W, W / 2 + 1.3, H / 2 - 0.7, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), cam.fx, cam.fy, cam.cx, cam.cy, W, H)
    r = _oracle(arrs, ocam, bg)
    gI = rs.standard_normal((3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    go = orc.render_backward(r, gI, gD)
    out = _hip_render(arrs, cam, W, H, bg, grads=(gI, gD))
    assert rel_to_max(out["image"], r.image) <= TOL and rel_to_max(out["depth"], r.depth) <= TOL
    for k in ["positions", "scales", "rotations", "colors", "opacities"]:
        assert rel_to_max(out["grad_" + k], go[k]) <= TOL, k


def test_step_is_hip_graph_capturable_and_replays_bitwise():
    """The C ABI never allocates, never synchronises and enqueues everything on the caller's stream (include/fgs.h),
    so forward + backward can be captured into a HIP graph and replayed: the replay must reproduce the eager
    gradients bit for bit, also after the inputs changed in place."""
    from fresnel_amd.renderer import Camera, TileBasedRenderer
    dev = _cuda()
    N, S, Bn = 4000, 128, 3
    rs = np.random.RandomState(17)
    per = [synth_aniso(N, 80 + b) for b in range(Bn)]
    leaves = [torch.from_numpy(np.stack([p[i] for p in per])).to(dev).requires_grad_(True) for i in range(5)]
    cam = Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
    ren = TileBasedRenderer(S, S, background=(0.1, 0.2, 0.3))
    gI = torch.from_numpy(rs.standard_normal((Bn, 3, S, S)).astype(np.float32)).to(dev)
    gD = torch.from_numpy((rs.standard_normal((Bn, S, S)) * 0.1).astype(np.float32)).to(dev)
    outs = [torch.zeros_like(t) for t in leaves]

    def step():
        img, dep = ren(*leaves, cam, return_depth=True)
        for g, o in zip(torch.autograd.grad([img, dep], leaves, [gI, gD]), outs):
            o.copy_(g)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for trial in range(2):
        step()
        eager = [o.clone() for o in outs]
        for o in outs:
            o.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(eager, outs):
            assert torch.equal(a, b)
        with torch.no_grad():  # new inputs in the same buffers: the graph must follow them
            leaves[0].add_(0.01)
            leaves[4].mul_(0.9)
