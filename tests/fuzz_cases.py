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
