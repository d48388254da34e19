"""One-off: the fuzz_asm seed-3 / iteration-10 case (phases gradient 3.2e-4 off the fp32 torch oracle): HIP vs the
oracle in fp32 and in fp64 -- who is off?"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import asm_oracle, fgs_oracle as orc
from fresnel_amd.renderer import Camera, ASMWaveFieldRenderer
dev = torch.device('cuda:0')
rs = np.random.RandomState(3)
for it in range(11):
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
    if it % 2 == 0:
        P = int(rs.choice([1, 4, 16])); wl = np.array([0.07, 0.052, 0.043], np.float32) * float(rs.uniform(0.8, 1.3))
        kw = dict(num_depth_planes=P, depth_range=(0.1, 3.2), focal_depth=float(rs.uniform(0.3, 1.5)), pixel_pitch=1.0 / float(rs.choice([128, 256])))
    if it != 10:
        continue
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
    ph = torch.from_numpy(phases).to(dev).requires_grad_(True)
    ren = ASMWaveFieldRenderer(W, H, background=bg, **kw).to(dev)
    img = ren(*ts, cam, phases=ph, wavelengths_rgb=torch.from_numpy(wl).to(dev))
    (img * torch.from_numpy(gI).to(dev)).sum().backward()
    args = dict(bg=bg, num_planes=P, depth_range=(0.1, 3.2), focal_depth=kw['focal_depth'], pixel_pitch=kw['pixel_pitch'], grad_out=gI)
    r32 = asm_oracle.render(*arrs, phases, wl, ocam, **args)
    r64 = asm_oracle.render(*arrs, phases, wl, ocam, dtype=torch.float64, **args)
    for t, k in zip(ts + [ph], ["positions", "scales", "rotations", "colors", "opacities", "phases"]):
        g = t.grad.cpu().numpy()
        print(k, 'hip vs o32 %.2e' % rel_to_max(g, r32['grad_' + k]), ' hip vs o64 %.2e' % rel_to_max(g, r64['grad_' + k]),
              ' o32 vs o64 %.2e' % rel_to_max(r32['grad_' + k], r64['grad_' + k]))
    print('image hip-o32 %.2e hip-o64 %.2e o32-o64 %.2e' % (np.abs(img.detach().cpu().numpy() - r32['image']).max(),
          np.abs(img.detach().cpu().numpy() - r64['image']).max(), np.abs(r32['image'] - r64['image']).max()))
