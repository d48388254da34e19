"""Soak: many steps of every path; memory must stay flat and results finite (not a test).
python scratch/fuzz/soak.py [commit]  -- every run prints its elapsed seconds; the header carries fgs_version() and the commit."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
import bench; bench._import_compute()
from fresnel_amd import renderer as R
dev = torch.device('cuda:0')
from fresnel_amd import _binding as _B
print(f"# soak, library: {_B.version()}   commit: {sys.argv[1] if len(sys.argv) > 1 else 'unknown'}   date: {time.strftime('%Y-%m-%d %H:%M:%S')}", flush=True)
_T0 = [time.time()]
def _lap():
    t = time.time() - _T0[0]; _T0[0] = time.time(); return t
def run(tag, n_img, N, S, steps, use_phase=False, skip=False):
    pos, scale, quat, col, opa = bench.synth_batch(n_img, N, 77, dev)
    leaves = [t.requires_grad_(True) for t in (pos, scale, quat, col, opa)]
    ph = torch.rand(n_img, N, device=dev).requires_grad_(True) if use_phase else None
    cam_t = R.pack_cameras(R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S), dev)
    cfg = R._Cfg(S, S, (0.1, 0.2, 0.3), 64, use_phase, 0.25, saturation_skip=skip)
    gI = torch.randn(n_img, 3, S, S, device=dev); gD = torch.randn(n_img, S, S, device=dev) * 0.1
    m0 = None
    for i in range(steps):
        for t in leaves: t.grad = None
        img, dep = R.GaussianRenderer.apply(*leaves, ph, cam_t, cfg)
        torch.autograd.backward([img, dep], [gI, gD])
        if i == 10: torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
    torch.cuda.synchronize()
    ok = all(torch.isfinite(t.grad).all().item() for t in leaves) and torch.isfinite(img).all().item()
    print(tag, 'steps', steps, 'finite', ok, 'mem delta MB', (torch.cuda.memory_allocated() - m0) / 1e6, 'peak GB', torch.cuda.max_memory_allocated() / 1e9, 'seconds %.1f' % _lap(), flush=True)
run('config3', 8, 32768, 512, 300)
run('config3 skip', 8, 32768, 512, 300, skip=True)
run('config2', 16, 8192, 256, 300)
run('config4 phase', 16, 8192, 256, 200, use_phase=True)
run('B=32', 32, 32768, 512, 30)


def run_asm(tag, n_img, N, S, steps):
    from fresnel_amd.renderer import ASMWaveFieldRenderer, Camera
    pos, scale, quat, col, opa = bench.synth_batch(n_img, N, 78, dev)
    leaves = [t.requires_grad_(True) for t in (pos, scale, quat, col, opa)]
    ph = (torch.rand(n_img, N, 3, device=dev) * 6.28).requires_grad_(True)
    wl = torch.tensor([0.0635, 0.05, 0.041], device=dev, requires_grad=True)
    ren = ASMWaveFieldRenderer(S, S).to(dev)
    cam = Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
    gI = torch.randn(n_img, 3, S, S, device=dev)
    m0 = None
    for i in range(steps):
        for t in leaves + [ph, wl]: t.grad = None
        img = ren(*leaves, cam, phases=ph, wavelengths_rgb=wl)
        img.backward(gI)
        if i == 10: torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
    torch.cuda.synchronize()
    ok = all(torch.isfinite(t.grad).all().item() for t in leaves + [ph, wl]) and torch.isfinite(img).all().item()
    print(tag, 'steps', steps, 'finite', ok, 'mem delta MB', (torch.cuda.memory_allocated() - m0) / 1e6, 'peak GB', torch.cuda.max_memory_allocated() / 1e9, 'seconds %.1f' % _lap(), flush=True)


run_asm('config5 b1', 1, 32768, 512, 200)
run_asm('config5 b8', 8, 32768, 512, 60)
run_asm('asm 256 b4', 4, 8192, 256, 100)
