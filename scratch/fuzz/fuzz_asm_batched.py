"""Randomized parity sweep of the BATCHED angular-spectrum renderer on column-kernel shapes (power-of-two heights, whole column
tiles) against the torch oracle: fuzz_asm_batched.py [seed] [cases].  Not a test; prints every case and the worst tensor."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import asm_oracle, fgs_oracle as orc
from fresnel_amd.renderer import Camera
import test_hip_asm as T
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 12
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1  # run just this iteration (the random stream is consumed alike)
rs = np.random.RandomState(1000 + seed)
worst = 0.0
for it in range(ncase):
    H = int(rs.choice([64, 64, 128, 256])); W = 16 * int(rs.randint(4, 26)); P = int(rs.choice([2, 5, 6, 9, 16])); Bn = int(rs.choice([1, 2, 3, 5]))
    N = int(rs.choice([40, 150, 300]))
    near, far = 0.3, float(rs.uniform(1.5, 3.0))
    bg = tuple(float(x) for x in rs.rand(3) * 0.3)
    per = []
    for b in range(Bn):
        a = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=0.9, smin=0.03, smax=float(rs.choice([0.06, 0.12]))))
        if H > W: a[0][:, 1] *= H / W * 0.6
        lo = float(rs.uniform(near, far - 0.2)); hi = float(rs.uniform(lo + 0.1, far))  # a depth band: some planes stay empty
        a[0][:, 2] = -rs.uniform(lo, hi, N).astype(np.float32)
        per.append(a)
    arrs = [np.stack([p[i] for p in per]) for i in range(5)]
    phases = (rs.random_sample((Bn, N, 3)) * 2 * np.pi).astype(np.float32)
    wl = np.array([0.07, 0.052, 0.043], np.float32) * float(rs.uniform(0.8, 1.3))
    gI = rs.standard_normal((Bn, 3, H, W)).astype(np.float32)
    f = 0.8 * min(W, H)
    cam = Camera(f, f, W / 2, H / 2, W, H)
    kw = dict(num_depth_planes=P, depth_range=(near, far), focal_depth=float(rs.uniform(0.5, 1.5)), pixel_pitch=1.0 / float(rs.choice([128, 200, 256])))
    if only >= 0 and it != only: continue
    out = T._hip_asm(arrs, phases, wl, cam, W, H, bg, gI=gI, **kw)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), f, f, W / 2, H / 2, W, H)
    errs, gw = {}, 0.0
    for b in range(Bn):
        r = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=bg, num_planes=P, depth_range=(near, far), focal_depth=kw["focal_depth"],
                              pixel_pitch=kw["pixel_pitch"], grad_out=gI[b])
        errs["image%d" % b] = float(np.abs(out["image"][b] - r["image"]).max())
        for k in ["positions", "scales", "rotations", "colors", "opacities", "phases"]:
            errs["%s%d" % (k[:3], b)] = rel_to_max(out["grad_" + k][b], r["grad_" + k])
        gw = gw + r["grad_wavelengths"]
    errs["wavelengths"] = rel_to_max(out["grad_wavelengths"], gw)  # (1e-4 like the others since round 4)
    m = max(errs.values()); worst = max(worst, m)
    if only >= 0: print({k: '%.1e' % v for k, v in errs.items()})
    print(f"seed {seed} it {it:2d} W{W} H{H} P{P} B{Bn} N{N} max err {m:.2e} ({max(errs, key=errs.get)})" + ('' if m <= 1e-4 else '  <-- ABOVE 1e-4'), flush=True)
print('worst', worst)
