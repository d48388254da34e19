"""One-off randomized parity sweep: HIP vs the C oracle over odd shapes / radii / backgrounds (not a test)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import Camera, TileBasedRenderer
dev = torch.device('cuda:0')
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(40):
    W, H = int(rs.randint(5, 200)), int(rs.randint(5, 150))
    N = int(rs.choice([1, 3, 17, 63, 64, 65, 200, 900, 2500]))
    maxr = float(rs.choice([8, 20, 64, 150]))
    smax = float(rs.choice([0.02, 0.1, 0.4]))
    arrs = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=float(rs.choice([0.5, 1.0, 1.3])), smax=smax))
    bg = tuple(float(x) for x in rs.rand(3))
    fx = float(rs.uniform(0.5, 1.5) * W)
    cam = Camera(fx, fx, W / 2 + rs.uniform(-3, 3), H / 2 + rs.uniform(-3, 3), W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), cam.fx, cam.fy, cam.cx, cam.cy, W, H)
    r = orc.render(*arrs, ocam, bg=bg, max_radius=maxr)
    gI = rs.standard_normal((3, H, W)).astype(np.float32); gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    go = orc.render_backward(r, gI, gD)
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
    ren = TileBasedRenderer(W, H, background=bg, max_radius=maxr)
    ren.tuning = dict(tile_w=int(rs.choice([16, 32])))
    img, dep = ren(*ts, cam, return_depth=True)
    ((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
    errs = dict(image=rel_to_max(img.detach().cpu().numpy(), r.image), depth=rel_to_max(dep.detach().cpu().numpy(), r.depth))
    for t, k in zip(ts, ["positions", "scales", "rotations", "colors", "opacities"]):
        errs[k] = rel_to_max(t.grad.cpu().numpy(), go[k])
    m = max(errs.values()); worst = max(worst, m)
    flag = '' if m <= 1e-4 else '  <-- FAIL'
    print(f"it {it:2d} W{W} H{H} N{N} maxr{maxr} smax{smax} max err {m:.2e}{flag}", flush=True)
print('worst', worst)
