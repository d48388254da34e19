"""One-off randomized sweep: batched renders with one orbit camera per image, points around the origin (some
behind the camera / huge / tiny), vs the C oracle per image (not a test)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import TileBasedRenderer, create_camera_from_pose
dev = torch.device('cuda:0')
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(16):
    S = int(rs.choice([32, 64, 100, 144])); Bn = int(rs.choice([1, 2, 5])); N = int(rs.choice([40, 300, 1500]))
    pos = (rs.standard_normal((Bn, N, 3)) * float(rs.choice([0.3, 1.0, 2.5]))).astype(np.float32)  # some land behind the camera
    smin, smax = [(0.003, 1.5), (0.01, 0.3), (0.02, 0.15)][int(sys.argv[2]) if len(sys.argv) > 2 else 0]
    scale = np.exp(rs.uniform(np.log(smin), np.log(smax), (Bn, N, 3))).astype(np.float32)
    quat = rs.standard_normal((Bn, N, 4)).astype(np.float32)
    col = rs.rand(Bn, N, 3).astype(np.float32); opa = rs.uniform(0.0, 1.1, (Bn, N)).astype(np.float32)
    cams = [create_camera_from_pose(float(rs.uniform(-1.2, 1.2)), float(rs.uniform(0, 6.28)), S, distance=float(rs.uniform(1.0, 4.0))) for _ in range(Bn)]
    bg = tuple(float(x) for x in rs.rand(3))
    gI = rs.standard_normal((Bn, 3, S, S)).astype(np.float32); gD = (rs.standard_normal((Bn, S, S)) * 0.1).astype(np.float32)
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (pos, scale, quat, col, opa)]
    img, dep = TileBasedRenderer(S, S, background=bg)(*ts, cams, return_depth=True)
    ((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
    m = 0.0; where = ''
    for b in range(Bn):
        c = cams[b]
        ocam = orc.make_camera(c.view_matrix.numpy(), c.fx, c.fy, c.cx, c.cy, S, S)
        r = orc.render(pos[b], scale[b], quat[b], col[b], opa[b], ocam, bg=bg)
        go = orc.render_backward(r, gI[b], gD[b])
        errs = dict(image=rel_to_max(img[b].detach().cpu().numpy(), r.image), depth=rel_to_max(dep[b].detach().cpu().numpy(), r.depth))
        for t, k in zip(ts, ["positions", "scales", "rotations", "colors", "opacities"]):
            errs[k] = rel_to_max(t.grad[b].cpu().numpy(), go[k])
        if max(errs.values()) > m: m = max(errs.values()); where = f'img{b}:' + max(errs, key=errs.get)
    worst = max(worst, m)
    print(f"it {it:2d} S{S} B{Bn} N{N} max err {m:.2e} ({where})" + ('' if m <= 1e-4 else '  <-- FAIL'), flush=True)
print('worst', worst)
