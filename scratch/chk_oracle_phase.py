import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import synth_aniso, rel_to_max
from oracle import fgs_oracle as orc
torch.set_num_threads(8)
W, H, N = 144, 112, 200
rs = np.random.RandomState(3)
a = list(synth_aniso(N, 50, opacity_max=1.0, smin=0.02, smax=0.09))
phases = rs.random_sample(N).astype(np.float32)
ocam = orc.make_camera(np.eye(4, dtype=np.float32), 0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
gI = rs.standard_normal((3, H, W)).astype(np.float32); gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
bg = (0.05, 0.1, 0.15); amp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
if len(sys.argv) > 3: exec(sys.argv[3])
if len(sys.argv) > 3: exec(sys.argv[3])
r = orc.render(*a, ocam, bg=bg, phases=phases, phase_amp=amp)
go = orc.render_backward(r, gI, gD)
# torch out-of-place restatement of the composite (float64 to act as ground truth)
dt = torch.float32 if len(sys.argv) > 1 else torch.float64
mean = torch.tensor(r.proj['mean2d'], dtype=dt, requires_grad=True)
conic = torch.tensor(r.proj['conic'], dtype=dt, requires_grad=True)
opa = torch.tensor(a[4], dtype=dt, requires_grad=True)
col = torch.tensor(a[3], dtype=dt, requires_grad=True)
dep = torch.tensor(r.proj['depth'], dtype=dt, requires_grad=True)
ph = torch.tensor(phases, dtype=dt, requires_grad=True)
C = torch.zeros(H, W, 3, dtype=dt); A = torch.zeros(H, W, dtype=dt); D = torch.zeros(H, W, dtype=dt); P = torch.zeros(H, W, dtype=dt)
for i in r.vis_sorted.tolist():
    x0, x1, y0, y1 = [int(t) for t in r.proj['bbox'][i]]
    if x0 >= x1 or y0 >= y1: continue
    ly, lx = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing='ij')
    dx = lx - mean[i, 0]; dy = ly - mean[i, 1]
    m = conic[i, 0] * dx * dx + conic[i, 1] * dx * dy + conic[i, 2] * dy * dy
    alpha = torch.exp(-0.5 * m) * opa[i]
    pd = torch.abs(ph[i] - P[y0:y1, x0:x1]); pd = torch.min(pd, 1.0 - pd)
    alpha = alpha * ((1 - amp) + amp * torch.cos(pd * 2 * 3.14159))
    alpha = torch.clamp(alpha, 0, 0.99)
    w = alpha * (1 - A[y0:y1, x0:x1])
    mask = torch.zeros(H, W, dtype=torch.bool); mask[y0:y1, x0:x1] = True
    wf = torch.zeros(H, W, dtype=dt).masked_scatter(mask, w)
    C = C + wf.unsqueeze(-1) * col[i].view(1, 1, 3); D = D + wf * dep[i]; A = A + wf
    pc = wf / A.clamp(min=1e-6)
    P = torch.where(mask, P * (1 - pc) + ph[i] * pc, P)
C = C + (1 - A).unsqueeze(-1) * torch.tensor(bg, dtype=dt).view(1, 1, 3)
img = torch.clamp(C.permute(2, 0, 1), 0, 1)
print('fwd img diff', float((img.detach().float() - torch.from_numpy(r.image)).abs().max()))
((img * torch.tensor(gI, dtype=dt)).sum() + (D * torch.tensor(gD, dtype=dt)).sum()).backward()
for k, t in [('mean2d', mean), ('conic', conic), ('opacities', opa), ('colors', col), ('depth', dep), ('phases', ph)]:
    print(k, 'oracle vs torch64 rel', rel_to_max(go[k], t.grad.numpy()))
