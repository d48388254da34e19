import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import synth_aniso, rel_to_max
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import Camera, TileBasedRenderer
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
dev = torch.device('cuda:0')
for use_phase in (True, False):
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
    ph = torch.from_numpy(phases).to(dev).requires_grad_(True)
    ren = TileBasedRenderer(W, H, background=(0.05, 0.1, 0.15), use_phase_blending=use_phase)
    img, dep = ren(*ts, cam, return_depth=True, phases=ph)
    ((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
    for b in range(Bn):
        r = orc.render([a[b] for a in arrs][0], *[a[b] for a in arrs][1:], ocam, bg=(0.05, 0.1, 0.15), phases=phases[b] if use_phase else None, phase_amp=0.25)
        go = orc.render_backward(r, gI[b], gD[b])
        print('phase', use_phase, 'b', b, 'img', rel_to_max(img[b].detach().cpu().numpy(), r.image))
        for k, t in zip(["positions", "scales", "rotations", "colors", "opacities"], ts):
            g = t.grad[b].cpu().numpy()
            err = np.abs(g - go[k]); mx = np.abs(go[k]).max()
            idx = np.argsort(err.reshape(len(g), -1).max(1))[::-1][:4]
            print('   ', k, 'rel', err.max() / mx, 'top idx', idx, 'errs', err.reshape(len(g), -1).max(1)[idx] / mx)
        if use_phase:
            g = ph.grad[b].cpu().numpy(); err = np.abs(g - go['phases']); print('    phases rel', err.max() / np.abs(go['phases']).max())
