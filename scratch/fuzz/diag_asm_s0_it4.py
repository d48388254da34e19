"""Diagnosis of sweep case asm s0 it 4 (one Gaussian's quaternion gradient 1.1e-4 from the oracle): the SAME scene through the
wave-field renderer (same splat kernels and projection adjoint, no FFT chain) against the torch oracle, per Gaussian."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import fuzz_cases as FC
from oracle import asm_oracle, fgs_oracle as orc
from fresnel_amd.renderer import Camera, WaveFieldRenderer, ASMWaveFieldRenderer
dev = torch.device('cuda:0')
c = [c for c in FC.asm_cases(0) if c['it'] == 4][0]
W, H = c['W'], c['H']
cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
ocam = orc.make_camera(np.eye(4), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def hip_wave(sel=None):
    arrs = [a if sel is None else a[sel] for a in c['arrs']]; ph = c['phases'] if sel is None else c['phases'][sel]
    ts = [up(a).requires_grad_(True) for a in arrs]; pht = up(ph).requires_grad_(True)
    img, dep = WaveFieldRenderer(W, H, background=c['bg']).to(dev)(*ts, cam, return_depth=True, phases=pht)
    ((img * up(c['gI'])).sum() + (dep * up(c['gD'])).sum()).backward()
    o = asm_oracle.render_wave(*arrs, ph, ocam, bg=c['bg'], grad_out=c['gI'], grad_depth=c['gD'])
    o64 = asm_oracle.render_wave(*arrs, ph, ocam, bg=c['bg'], grad_out=c['gI'], grad_depth=c['gD'], dtype=torch.float64, project_f64=True)
    return [t.grad.cpu().numpy() for t in ts], o, o64
g, o, o64 = hip_wave()
m = np.abs(o64['grad_rotations']).max()
d = np.abs(g[2] - o64['grad_rotations']).max(1) / m
print('WAVE renderer, whole scene: rotations worst %.2e at %d; G171 %.2e (oracle fp32 there %.2e)' % (d.max(), d.argmax(), d[171], np.abs(o['grad_rotations'][171] - o64['grad_rotations'][171]).max() / m))
for k, name in ((0, 'positions'), (1, 'scales'), (2, 'rotations')):
    mm = np.abs(o64['grad_' + name]).max()
    print('   ', name, 'hip-o64 %.2e  o32-o64 %.2e' % (np.abs(g[k] - o64['grad_' + name]).max() / mm, np.abs(o['grad_' + name] - o64['grad_' + name]).max() / mm))
sel = np.array([171])
g1, o1, o1_64 = hip_wave(sel)
m1 = np.abs(o1_64['grad_rotations']).max()
print('WAVE renderer, Gaussian 171 alone: rotations hip-o64 %.2e  o32-o64 %.2e   grad' % (np.abs(g1[2] - o1_64['grad_rotations']).max() / m1, np.abs(o1['grad_rotations'] - o1_64['grad_rotations']).max() / m1), o1_64['grad_rotations'][0], 'hip', g1[2][0])
