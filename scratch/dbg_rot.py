"""Which Gaussians carry the rotation-gradient mismatch of scratch/fuzz_batch.py?  (GPU box)"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max
from oracle import fgs_oracle as orc
from fresnel_amd.renderer import TileBasedRenderer, create_camera_from_pose
dev = torch.device('cuda:0')
rs = np.random.RandomState(7)
S, Bn, N = 144, 1, 300
# same draws as fuzz_batch.py it 0 of seed 7
_ = (rs.choice([32, 64, 100, 144]), rs.choice([1, 2, 5]), rs.choice([40, 300, 1500]))
pos = (rs.standard_normal((Bn, N, 3)) * float(rs.choice([0.3, 1.0, 2.5]))).astype(np.float32)
scale = np.exp(rs.uniform(np.log(0.003), np.log(1.5), (Bn, N, 3))).astype(np.float32)
quat = rs.standard_normal((Bn, N, 4)).astype(np.float32)
col = rs.rand(Bn, N, 3).astype(np.float32); opa = rs.uniform(0.0, 1.1, (Bn, N)).astype(np.float32)
cams = [create_camera_from_pose(float(rs.uniform(-1.2, 1.2)), float(rs.uniform(0, 6.28)), S, distance=float(rs.uniform(1.0, 4.0))) for _ in range(Bn)]
bg = tuple(float(x) for x in rs.rand(3))
gI = rs.standard_normal((Bn, 3, S, S)).astype(np.float32); gD = (rs.standard_normal((Bn, S, S)) * 0.1).astype(np.float32)
ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (pos, scale, quat, col, opa)]
img, dep = TileBasedRenderer(S, S, background=bg)(*ts, cams, return_depth=True)
((img * torch.from_numpy(gI).to(dev)).sum() + (dep * torch.from_numpy(gD).to(dev)).sum()).backward()
c = cams[0]
ocam = orc.make_camera(c.view_matrix.numpy(), c.fx, c.fy, c.cx, c.cy, S, S)
r = orc.render(pos[0], scale[0], quat[0], col[0], opa[0], ocam, bg=bg)
go = orc.render_backward(r, gI[0], gD[0])
gh = ts[2].grad[0].cpu().numpy(); gr = go['rotations']
d = np.abs(gh - gr).max(1); mx = np.abs(gr).max()
idx = np.argsort(-d)[:8]
print('max |grad rot|', mx)
for i in idx:
    print(i, 'diff %.3e' % d[i], 'hip', gh[i], 'orc', gr[i], 'scale', scale[0, i], '|q|', np.linalg.norm(quat[0, i]), 'z', r.proj['depth'][i] if 'depth' in r.proj else None, 'bbox', r.proj['bbox'][i])
print('scales grad rel', rel_to_max(ts[1].grad[0].cpu().numpy(), go['scales']), 'positions', rel_to_max(ts[0].grad[0].cpu().numpy(), go['positions']))
np.savez('gpurun_out/rotcase.npz', pos=pos[0], scale=scale[0], quat=quat[0], col=col[0], opa=opa[0], view=c.view_matrix.numpy(), S=S, bg=np.array(bg), gI=gI[0], gD=gD[0], hip_rot=gh, hip_scale=ts[1].grad[0].cpu().numpy(), hip_pos=ts[0].grad[0].cpu().numpy())
