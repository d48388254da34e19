"""CPU analysis of a dumped phase-path mismatch: oracle (fp32 C) vs fp64 autograd restatement vs HIP grads."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max
from oracle import fgs_oracle as orc
d = np.load(sys.argv[1])
W, H, amp, bg = int(d['W']), int(d['H']), float(d['amp']), tuple(float(x) for x in d['bg'])
a = [d[f'a{i}'] for i in range(5)]; phases = d['phases']; gI, gD = d['gI'], d['gD']
cam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
r = orc.render(*a, cam, bg=bg, phases=phases, phase_amp=amp)
go = orc.render_backward(r, gI, gD)
for k in ['positions', 'scales', 'rotations', 'colors', 'opacities', 'phases']:
    print('HIP vs oracle', k, rel_to_max(d['hip_' + k], go[k]))
dt = torch.float64
leaf = lambda x: torch.tensor(x, dtype=dt, requires_grad=True)
mean, conic, opa, col = leaf(r.proj["mean2d"]), leaf(r.proj["conic"]), leaf(a[4]), leaf(a[3])
dep, ph = leaf(r.proj["depth"]), leaf(phases)
C, A = torch.zeros(H, W, 3, dtype=dt), torch.zeros(H, W, dtype=dt)
D, P = torch.zeros(H, W, dtype=dt), torch.zeros(H, W, dtype=dt)
near = 0
for i in r.vis_sorted.tolist():
    x0, x1, y0, y1 = [int(t) for t in r.proj["bbox"][i]]
    if x0 >= x1 or y0 >= y1: continue
    ly, lx = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
    dx, dy = lx - mean[i, 0], ly - mean[i, 1]
    m = conic[i, 0] * dx * dx + conic[i, 1] * dx * dy + conic[i, 2] * dy * dy
    alpha = torch.exp(-0.5 * m) * opa[i]
    pd0 = torch.abs(ph[i] - P[y0:y1, x0:x1])
    near += int(((pd0 - 0.5).abs() < 1e-5).sum()) + int((pd0 < 1e-5).sum())
    pd = torch.min(pd0, 1.0 - pd0)
    raw = alpha * ((1 - amp) + amp * torch.cos(pd * 2 * 3.14159))
    near += int(((raw - 0.99).abs() < 1e-5).sum())
    alpha = torch.clamp(raw, 0, 0.99)
    w = alpha * (1 - A[y0:y1, x0:x1])
    mask = torch.zeros(H, W, dtype=torch.bool); mask[y0:y1, x0:x1] = True
    wf = torch.zeros(H, W, dtype=dt).masked_scatter(mask, w)
    C, D, A = C + wf.unsqueeze(-1) * col[i].view(1, 1, 3), D + wf * dep[i], A + wf
    pc = wf / A.clamp(min=1e-6)
    P = torch.where(mask, P * (1 - pc) + ph[i] * pc, P)
print('pixels within 1e-5 of a kink (|dphi|=0, 0.5, raw=0.99):', near)
C = C + (1 - A).unsqueeze(-1) * torch.tensor(bg, dtype=dt).view(1, 1, 3)
img = torch.clamp(C.permute(2, 0, 1), 0, 1)
print('image oracle vs fp64', float((img.detach().float() - torch.from_numpy(r.image)).abs().max()), ' HIP vs fp64', float((img.detach().float() - torch.from_numpy(d['hip_image'])).abs().max()))
((img * torch.tensor(gI, dtype=dt)).sum() + (D * torch.tensor(gD, dtype=dt)).sum()).backward()
for k, t in [("mean2d", mean), ("conic", conic), ("opacities", opa), ("colors", col), ("depth", dep), ("phases", ph)]:
    print('oracle vs fp64 (composite level)', k, rel_to_max(go[k], t.grad.numpy()))
print('HIP vs fp64 opacities', rel_to_max(d['hip_opacities'], opa.grad.numpy()), 'colors', rel_to_max(d['hip_colors'], col.grad.numpy()), 'phases', rel_to_max(d['hip_phases'], ph.grad.numpy()))
