"""One-off randomized parity sweep of the ASM and wave renderers vs the torch oracle (not a test)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import asm_oracle, fgs_oracle as orc
from fresnel_amd.renderer import Camera, ASMWaveFieldRenderer, WaveFieldRenderer
dev = torch.device('cuda:0')
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(14):
    W, H = int(rs.choice([32, 48, 64, 96, 120])), int(rs.choice([32, 40, 64, 88]))
    N = int(rs.choice([1, 17, 64, 200, 700]))
    rgbph = bool(rs.rand() < 0.5)
    arrs = list(synth_aniso(N, int(rs.randint(1 << 30)), opacity_max=0.9, smin=0.02, smax=float(rs.choice([0.05, 0.15]))))
    arrs[0][:, 2] = -rs.uniform(0.3, 3.0, N).astype(np.float32)
    phases = (rs.random_sample((N, 3) if rgbph else (N,)) * 2 * np.pi).astype(np.float32)
    bg = tuple(float(x) for x in rs.rand(3) * 0.3)
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), cam.fx, cam.fy, cam.cx, cam.cy, W, H)
    gI = rs.standard_normal((3, H, W)).astype(np.float32); gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
    ph = torch.from_numpy(phases).to(dev).requires_grad_(True)
    if it % 2 == 0:
        P = int(rs.choice([1, 4, 16])); wl = np.array([0.07, 0.052, 0.043], np.float32) * float(rs.uniform(0.8, 1.3))
        kw = dict(num_depth_planes=P, depth_range=(0.1, 3.2), focal_depth=float(rs.uniform(0.3, 1.5)), pixel_pitch=1.0 / float(rs.choice([128, 256])))
        ren = ASMWaveFieldRenderer(W, H, background=bg, **kw).to(dev)
        wlt = torch.from_numpy(wl).to(dev)
        img = ren(*ts, cam, phases=ph, wavelengths_rgb=wlt)
        r = asm_oracle.render(*arrs, phases, wl, ocam, bg=bg, num_planes=P, depth_range=(0.1, 3.2), focal_depth=kw['focal_depth'], pixel_pitch=kw['pixel_pitch'], grad_out=gI)
        (img * torch.from_numpy(gI).to(dev)).sum().backward()
        errs = dict(image=float(np.abs(img.detach().cpu().numpy() - r['image']).max()))
        tag = f'ASM P{P}'
    else:
        ren = WaveFieldRenderer(W, H, background=bg).to(dev)
        img, dep = ren(*ts, cam, return_depth=True, phases=ph)
        r = asm_oracle.render_wave(*arrs, phases, ocam, bg=bg, grad_out=gI, grad_depth=gD)
        ((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
        errs = dict(image=float(np.abs(img.detach().cpu().numpy() - r['image']).max()), depth=rel_to_max(dep.detach().cpu().numpy(), r['depth']))
        tag = 'WAVE'
    for t, k in zip(ts + [ph], ["positions", "scales", "rotations", "colors", "opacities", "phases"]):
        if k == "phases" and N == 1:
            continue  # a single Gaussian's phase is a global phase: the true gradient is 0, the ratio is noise / noise
        errs[k] = rel_to_max(t.grad.cpu().numpy(), r['grad_' + k])
    m = max(errs.values()); worst = max(worst, m)
    print(f"it {it:2d} {tag} W{W} H{H} N{N} rgbph{int(rgbph)} max err {m:.2e} ({max(errs, key=errs.get)})" + ('' if m <= 1e-4 else '  <-- FAIL'), flush=True)
print('worst', worst)
