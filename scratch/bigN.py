"""One-off: N = 300 000 Gaussians in one 512x512 image vs the C oracle (long lists: ~8000 entries per tile)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_saag
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import Camera, TileBasedRenderer
dev = torch.device('cuda:0')
N, S = 300000, 512
arrs = list(synth_saag(N, 5)); arrs[4][:] = 0.02  # low opacity: the deep part of the lists still matters
rs = np.random.RandomState(1)
cam = Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
ocam = orc.make_camera(np.eye(4, dtype=np.float32), cam.fx, cam.fy, cam.cx, cam.cy, S, S)
gI = rs.standard_normal((3, S, S)).astype(np.float32); gD = (rs.standard_normal((S, S)) * 0.1).astype(np.float32)
ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
t0 = time.time()
img, dep = TileBasedRenderer(S, S)(*ts, cam, return_depth=True)
((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
torch.cuda.synchronize(); print('hip fwd+bwd s', time.time() - t0, flush=True)
t0 = time.time()
r = orc.render(*arrs, ocam)
go = orc.render_backward(r, gI, gD)
print('oracle s', time.time() - t0, 'P', r.P, flush=True)
errs = dict(image=rel_to_max(img.detach().cpu().numpy(), r.image), depth=rel_to_max(dep.detach().cpu().numpy(), r.depth))
for t, k in zip(ts, ["positions", "scales", "rotations", "colors", "opacities"]):
    errs[k] = rel_to_max(t.grad.cpu().numpy(), go[k])
print(errs)
