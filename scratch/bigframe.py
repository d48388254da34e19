import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import Camera, TileBasedRenderer
dev = torch.device('cuda:0')
rs = np.random.RandomState(3)
W, H, N = 1200, 1100, 3000
arrs = list(synth_aniso(N, 5, smax=0.08)); bg = (0.1, 0.2, 0.3)
cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
ocam = orc.make_camera(np.eye(4, dtype=np.float32), cam.fx, cam.fy, cam.cx, cam.cy, W, H)
r = orc.render(*arrs, ocam, bg=bg)
gI = rs.standard_normal((3, H, W)).astype(np.float32); gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
go = orc.render_backward(r, gI, gD)
ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
img, dep = TileBasedRenderer(W, H, background=bg)(*ts, cam, return_depth=True)
((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
errs = dict(image=rel_to_max(img.detach().cpu().numpy(), r.image))
for t, k in zip(ts, ["positions", "scales", "rotations", "colors", "opacities"]):
    errs[k] = rel_to_max(t.grad.cpu().numpy(), go[k])
print('tiles', ((W + 15) // 16) * ((H + 15) // 16), 'P', r.P, errs)
