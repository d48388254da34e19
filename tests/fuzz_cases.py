"""Seeded case generators of the randomized parity sweeps (scratch/fuzz/fuzz_phase.py, scratch/fuzz/fuzz_asm.py).  Kept in one
place so that a sweep's failing case (seed, iteration) can be turned into a committed fixture: tests/golden/
make_goldens.py --kinks-only replays the same draws, runs the reference (fp32 and fp64) on them and stores the result.
The draw ORDER is part of the contract -- do not reorder statements."""
import numpy as np

from helpers import synth_aniso


def phase_cases(seed, n_iter=24):
    """Phase-blending path (TileBasedRenderer(use_phase_blending=True)): odd frame sizes, tiny to mid N, amplitudes
    < 0.5 (the interference factor stays positive), half the cases with zone-quantised depths (config 4)."""
    rs = np.random.RandomState(seed)
    for it in range(n_iter):
        W, H = int(rs.randint(5, 160)), int(rs.randint(5, 120))
        N = int(rs.choice([1, 17, 64, 65, 200, 900, 2000]))
        amp = float(rs.choice([0.1, 0.25, 0.45]))
        arrs = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=float(rs.choice([0.5, 1.0, 1.3])),
                                smax=float(rs.choice([0.03, 0.1, 0.3]))))
        if rs.rand() < 0.5:
            arrs[0][:, 2] = -2.0 - 2.0 * (np.floor(rs.rand(N) * 8) + 0.5) / 8
        phases = rs.rand(N).astype(np.float32)
        bg = tuple(float(x) for x in rs.rand(3))
        # (the camera draws nothing)
        gI = rs.standard_normal((3, H, W)).astype(np.float32)
        gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
        yield dict(it=it, W=W, H=H, N=N, amp=amp, arrs=arrs, phases=phases, bg=bg, gI=gI, gD=gD)


def asm_cases(seed, n_iter=14):
    """ASM (even iterations) and wave-field (odd) renderers on small frames."""
    rs = np.random.RandomState(seed)
    kw = None
    for it in range(n_iter):
        W, H = int(rs.choice([32, 48, 64, 96, 120])), int(rs.choice([32, 40, 64, 88]))
        N = int(rs.choice([1, 17, 64, 200, 700]))
        rgbph = bool(rs.rand() < 0.5)
        arrs = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=0.9, smin=0.02, smax=float(rs.choice([0.05, 0.15]))))
        arrs[0][:, 2] = -rs.uniform(0.3, 3.0, N).astype(np.float32)
        phases = (rs.random_sample((N, 3) if rgbph else (N,)) * 2 * np.pi).astype(np.float32)
        bg = tuple(float(x) for x in rs.rand(3) * 0.3)
        gI = rs.standard_normal((3, H, W)).astype(np.float32)
        gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
        case = dict(it=it, W=W, H=H, N=N, rgbph=rgbph, arrs=arrs, phases=phases, bg=bg, gI=gI, gD=gD, kind="wave")
        if it % 2 == 0:
            P = int(rs.choice([1, 4, 16]))
            wl = np.array([0.07, 0.052, 0.043], np.float32) * float(rs.uniform(0.8, 1.3))
            kw = dict(num_depth_planes=P, depth_range=(0.1, 3.2), focal_depth=float(rs.uniform(0.3, 1.5)),
                      pixel_pitch=1.0 / float(rs.choice([128, 256])))
            case.update(kind="asm", P=P, wl=wl, kw=kw)
        yield case


def blend_cases(seed, n_iter=40):
    """Blend path (TileBasedRenderer): odd frame sizes, N = 1 ... 2500, radius caps 8 ... 150, off-centre principal points, random
    backgrounds, 16- and 32-wide tiles (the draws of round 2-4's scratch/fuzz/fuzz.py, in its order)."""
    rs = np.random.RandomState(seed)
    for it in range(n_iter):
        W, H = int(rs.randint(5, 200)), int(rs.randint(5, 150))
        N = int(rs.choice([1, 3, 17, 63, 64, 65, 200, 900, 2500]))
        maxr = float(rs.choice([8, 20, 64, 150]))
        smax = float(rs.choice([0.02, 0.1, 0.4]))
        arrs = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=float(rs.choice([0.5, 1.0, 1.3])), smax=smax))
        bg = tuple(float(x) for x in rs.rand(3))
        fx = float(rs.uniform(0.5, 1.5) * W)
        cx = W / 2 + rs.uniform(-3, 3)
        cy = H / 2 + rs.uniform(-3, 3)
        gI = rs.standard_normal((3, H, W)).astype(np.float32)
        gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
        tile_w = int(rs.choice([16, 32]))
        yield dict(it=it, W=W, H=H, N=N, maxr=maxr, smax=smax, arrs=arrs, bg=bg, fx=fx, cx=float(cx), cy=float(cy), gI=gI, gD=gD,
                   tile_w=tile_w)


BATCH_SCALE_RANGES = [(0.003, 1.5), (0.01, 0.3), (0.02, 0.15)]


def batch_cases(seed, scale_range=2, n_iter=16):
    """Batched renders with one orbit camera per image, points around the origin (some behind the camera / huge / tiny); the
    camera of image b is create_camera_from_pose(el[b], az[b], S, distance=dist[b])."""
    rs = np.random.RandomState(seed)
    for it in range(n_iter):
        S = int(rs.choice([32, 64, 100, 144])); Bn = int(rs.choice([1, 2, 5])); N = int(rs.choice([40, 300, 1500]))
        pos = (rs.standard_normal((Bn, N, 3)) * float(rs.choice([0.3, 1.0, 2.5]))).astype(np.float32)
        smin, smax = BATCH_SCALE_RANGES[scale_range]
        scale = np.exp(rs.uniform(np.log(smin), np.log(smax), (Bn, N, 3))).astype(np.float32)
        quat = rs.standard_normal((Bn, N, 4)).astype(np.float32)
        col = rs.rand(Bn, N, 3).astype(np.float32); opa = rs.uniform(0.0, 1.1, (Bn, N)).astype(np.float32)
        poses = [(float(rs.uniform(-1.2, 1.2)), float(rs.uniform(0, 6.28)), float(rs.uniform(1.0, 4.0))) for _ in range(Bn)]
        bg = tuple(float(x) for x in rs.rand(3))
        gI = rs.standard_normal((Bn, 3, S, S)).astype(np.float32); gD = (rs.standard_normal((Bn, S, S)) * 0.1).astype(np.float32)
        yield dict(it=it, S=S, B=Bn, N=N, arrs=[pos, scale, quat, col, opa], poses=poses, bg=bg, gI=gI, gD=gD)


def asm_batched_cases(seed, n_iter=24):
    """Batched angular-spectrum renderer on column-kernel shapes (power-of-two heights, whole column tiles), depth bands that
    leave planes empty (the draws of round 3's scratch/fuzz/fuzz_asm_batched.py, in its order)."""
    rs = np.random.RandomState(1000 + seed)
    for it in range(n_iter):
        H = int(rs.choice([64, 64, 128, 256])); W = 16 * int(rs.randint(4, 26)); P = int(rs.choice([2, 5, 6, 9, 16])); Bn = int(rs.choice([1, 2, 3, 5]))
        N = int(rs.choice([40, 150, 300]))
        near, far = 0.3, float(rs.uniform(1.5, 3.0))
        bg = tuple(float(x) for x in rs.rand(3) * 0.3)
        per = []
        for b in range(Bn):
            a = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=0.9, smin=0.03, smax=float(rs.choice([0.06, 0.12]))))
            if H > W: a[0][:, 1] *= H / W * 0.6
            lo = float(rs.uniform(near, far - 0.2)); hi = float(rs.uniform(lo + 0.1, far))
            a[0][:, 2] = -rs.uniform(lo, hi, N).astype(np.float32)
            per.append(a)
        arrs = [np.stack([p[i] for p in per]) for i in range(5)]
        phases = (rs.random_sample((Bn, N, 3)) * 2 * np.pi).astype(np.float32)
        wl = np.array([0.07, 0.052, 0.043], np.float32) * float(rs.uniform(0.8, 1.3))
        gI = rs.standard_normal((Bn, 3, H, W)).astype(np.float32)
        f = 0.8 * min(W, H)
        kw = dict(num_depth_planes=P, depth_range=(near, far), focal_depth=float(rs.uniform(0.5, 1.5)),
                  pixel_pitch=1.0 / float(rs.choice([128, 200, 256])))
        yield dict(it=it, W=W, H=H, P=P, B=Bn, N=N, arrs=arrs, phases=phases, wl=wl, bg=bg, gI=gI, f=f, kw=kw)


def blend_big_cases(seed, n_iter=6):
    """Blend path at sizes where a tile's list spans several depth segments and list parts (round 5): frames of 200 ... 520 pixels,
    4 000 ... 30 000 Gaussians, automatic / 16- / 32-wide tiles, automatic / 64- / 128-entry segments, a random orbit camera."""
    rs = np.random.RandomState(7000 + seed)
    for it in range(n_iter):
        W, H = int(rs.randint(200, 521)), int(rs.randint(200, 521))
        N = int(rs.choice([4000, 12000, 30000]))
        smax = float(rs.choice([0.03, 0.08]))
        arrs = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=float(rs.choice([0.6, 1.0, 1.3])), smax=smax))
        bg = tuple(float(x) for x in rs.rand(3))
        fx = float(rs.uniform(0.6, 1.2) * W)
        gI = rs.standard_normal((3, H, W)).astype(np.float32)
        gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
        tuning = {}
        tw, sl = int(rs.choice([0, 16, 32])), int(rs.choice([0, 64, 128]))
        if tw: tuning["tile_w"] = tw
        if sl: tuning["seg_len"] = sl
        yield dict(it=it, W=W, H=H, N=N, maxr=64.0, smax=smax, arrs=arrs, bg=bg, fx=fx, cx=W / 2.0, cy=H / 2.0, gI=gI, gD=gD,
                   tuning=tuning, tile_w=tw, row_stride=4)
