"""One-off randomized parity sweep of the phase-blending path vs the C oracle (not a test)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import Camera, TileBasedRenderer
dev = torch.device('cuda:0')
from fuzz_cases import phase_cases
worst = 0.0
for c in phase_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 0):
    it, W, H, N, amp, arrs, phases, bg, gI, gD = (c[k] for k in ("it", "W", "H", "N", "amp", "arrs", "phases", "bg", "gI", "gD"))
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), cam.fx, cam.fy, cam.cx, cam.cy, W, H)
    r = orc.render(*arrs, ocam, bg=bg, phases=phases, phase_amp=amp)
    go = orc.render_backward(r, gI, gD)
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
    ph = torch.from_numpy(phases).to(dev).requires_grad_(True)
    ren = TileBasedRenderer(W, H, background=bg, use_phase_blending=True, phase_amplitude=amp)
    img, dep = ren(*ts, cam, return_depth=True, phases=ph)
    ((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
    errs = dict(image=rel_to_max(img.detach().cpu().numpy(), r.image), depth=rel_to_max(dep.detach().cpu().numpy(), r.depth))
    for t, k in zip(ts + [ph], ["positions", "scales", "rotations", "colors", "opacities", "phases"]):
        errs[k] = rel_to_max(t.grad.cpu().numpy(), go[k])
    m = max(errs.values()); worst = max(worst, m)
    if m > 1e-4:
        np.savez(f'gpurun_out/phasefail_{it}.npz', W=W, H=H, amp=amp, bg=np.array(bg), phases=phases, gI=gI, gD=gD, **{f'a{i}': a for i, a in enumerate(arrs)},
                 **{f'hip_{k}': t.grad.cpu().numpy() for t, k in zip(ts + [ph], ['positions', 'scales', 'rotations', 'colors', 'opacities', 'phases'])}, hip_image=img.detach().cpu().numpy())
    print(f"it {it:2d} W{W} H{H} N{N} amp{amp} max err {m:.2e} ({max(errs, key=errs.get)})" + ('' if m <= 1e-4 else '  <-- FAIL'), flush=True)
print('worst', worst)
