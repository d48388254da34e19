"""One-off randomized parity sweep of the ASM and wave renderers vs the torch oracle (not a test)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import asm_oracle, fgs_oracle as orc
from fresnel_amd.renderer import Camera, ASMWaveFieldRenderer, WaveFieldRenderer
dev = torch.device('cuda:0')
from fuzz_cases import asm_cases
worst = 0.0
for c in asm_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 0):
    it, W, H, N, rgbph, arrs, phases, bg, gI, gD = (c[k] for k in ("it", "W", "H", "N", "rgbph", "arrs", "phases", "bg", "gI", "gD"))
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), cam.fx, cam.fy, cam.cx, cam.cy, W, H)
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
    ph = torch.from_numpy(phases).to(dev).requires_grad_(True)
    if it % 2 == 0:
        P, wl, kw = c["P"], c["wl"], c["kw"]
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
