"""One-off: a few large / unusual frames vs the C oracle, default path and saturation_skip (not a test)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import Camera, TileBasedRenderer
dev = torch.device('cuda:0')
rs = np.random.RandomState(11)
for (W, H, N, smax) in [(1024, 768, 6000, 0.1), (2000, 40, 3000, 0.1), (16, 16, 5000, 0.2), (1537, 1, 500, 0.1), (640, 480, 20000, 0.05)]:
    arrs = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=1.0, smax=smax))
    bg = (0.1, 0.2, 0.3)
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), cam.fx, cam.fy, cam.cx, cam.cy, W, H)
    r = orc.render(*arrs, ocam, bg=bg)
    gI = rs.standard_normal((3, H, W)).astype(np.float32); gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    go = orc.render_backward(r, gI, gD)
    for skip, tw in ((False, 0), (False, 16), (False, 32), (True, 0)):
        ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
        ren = TileBasedRenderer(W, H, background=bg, saturation_skip=skip)
        ren.tuning = dict(tile_w=tw) if tw else None
        img, dep = ren(*ts, cam, return_depth=True)
        ((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
        errs = dict(image=rel_to_max(img.detach().cpu().numpy(), r.image), depth=rel_to_max(dep.detach().cpu().numpy(), r.depth))
        for t, k in zip(ts, ["positions", "scales", "rotations", "colors", "opacities"]):
            errs[k] = rel_to_max(t.grad.cpu().numpy(), go[k])
        m = max(errs.values())
        print(f"W{W} H{H} N{N} skip={skip} tile_w={tw} P={r.P} max err {m:.2e} ({max(errs, key=errs.get)})" + ('' if m <= 1e-4 else '  <-- FAIL'), flush=True)
