import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import synth_aniso, synth_saag, rel_to_max
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import Camera, TileBasedRenderer
from fresnel_amd import renderer as R
dev = torch.device('cuda:0')
def run(name, arrs, phases, W, H, bg=(0.05,0.1,0.15)):
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    rs = np.random.RandomState(1)
    gI = rs.standard_normal((3, H, W)).astype(np.float32); gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in arrs]
    ph = torch.from_numpy(phases).to(dev).requires_grad_(True)
    ren = TileBasedRenderer(W, H, background=bg, use_phase_blending=True)
    img, dep = ren(*ts, cam, return_depth=True, phases=ph)
    ((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
    r = orc.render(*arrs, ocam, bg=bg, phases=phases, phase_amp=0.25)
    go = orc.render_backward(r, gI, gD)
    rg, ids = orc.tile_lists(r.vis_sorted, r.proj['bbox'], W, H)
    print(name, 'maxlist', int(np.diff(rg).max()), 'img', rel_to_max(img.detach().cpu().numpy(), r.image),
          {k: float('%.2e' % rel_to_max(t.grad.cpu().numpy(), go[k])) for k, t in zip(["positions", "scales", "rotations", "colors", "opacities"], ts)},
          'phases', rel_to_max(ph.grad.cpu().numpy(), go['phases']))
rs = np.random.RandomState(3)
for N, S in [(200, 128), (600, 128), (2048, 128), (4096, 256)]:
    run(f'saag N={N}', list(synth_saag(N, 5)), rs.random_sample(N).astype(np.float32), S, S)
for N in (200, 1500):
    a = list(synth_aniso(N, 50, opacity_max=1.0, smin=0.02, smax=0.09))
    run(f'aniso N={N}', a, rs.random_sample(N).astype(np.float32), 144, 112)
    a[0][:, 2] = (-2.0 - 2.0 * (rs.randint(0, 8, N) + 0.5) / 8.0).astype(np.float32)
    run(f'aniso zones N={N}', a, rs.random_sample(N).astype(np.float32), 144, 112)
