import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import synth_aniso
from oracle import fgs_oracle as orc
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
ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
gI = rs.standard_normal((Bn, 3, H, W)).astype(np.float32)
gD = (rs.standard_normal((Bn, H, W)) * 0.1).astype(np.float32)
b = 1
a = [x[b] for x in arrs]
r = orc.render(*a, ocam, bg=(0.05, 0.1, 0.15), phases=phases[b], phase_amp=0.25)
g0 = orc.render_backward(r, gI[b], gD[b])
for eps in (1e-7, 1e-6):
    a2 = [x.copy() for x in a]
    a2[4] = (a2[4] * (1 + eps)).astype(np.float32)   # perturb opacities
    r2 = orc.render(*a2, ocam, bg=(0.05, 0.1, 0.15), phases=phases[b], phase_amp=0.25)
    g2 = orc.render_backward(r2, gI[b], gD[b])
    for k in ['positions', 'scales', 'rotations', 'opacities', 'phases']:
        err = np.abs(g2[k] - g0[k]).reshape(N, -1).max(1); mx = np.abs(g0[k]).max()
        i = np.argsort(err)[::-1][:3]
        print(eps, k, 'rel', err.max() / mx, i, err[i] / mx)
print('opacity of 815, 737:', a[4][815], a[4][737], 'phase', phases[b][815])
