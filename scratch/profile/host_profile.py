"""Where the HOST time of the per-image drop-in route goes: cProfile of B sequential (N,.) module calls +
torch.stack + backward at config 1's shape.  python scratch/profile/host_profile.py [N] [S] [B]"""
import cProfile, pstats, sys, time, io
sys.path.insert(0, '.')
import torch
import bench; bench._import_compute()
from fresnel_amd import renderer as R
N, S, Bn = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 256), (2, 128), (3, 32)))
dev = torch.device('cuda:0')
pos, scale, quat, col, opa = [t.requires_grad_(True) for t in bench.synth_batch(Bn, N, 1000, dev)]
cam = R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
ren = R.TileBasedRenderer(S, S).to(dev)
gI = torch.randn(Bn, 3, S, S, device=dev); gD = torch.randn(Bn, S, S, device=dev) * 0.1
def step():
    for t in (pos, scale, quat, col, opa): t.grad = None
    imgs, deps = [], []
    for b in range(Bn):
        im, dp = ren(pos[b], scale[b], quat[b], col[b], opa[b], cam, return_depth=True)
        imgs.append(im); deps.append(dp)
    torch.autograd.backward([torch.stack(imgs), torch.stack(deps)], [gI, gD])
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"N {N} S {S} B {Bn}: host {(t1 - t0) / 20 / Bn * 1e6:.1f} us per image, with the final sync {(t2 - t0) / 20 / Bn * 1e6:.1f} us per image")
# forward only, no autograd
with torch.no_grad():
    for _ in range(5):
        for b in range(Bn): ren(pos[b], scale[b], quat[b], col[b], opa[b], cam, return_depth=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        for b in range(Bn): ren(pos[b], scale[b], quat[b], col[b], opa[b], cam, return_depth=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"forward only, no_grad: host {(t1 - t0) / 20 / Bn * 1e6:.1f} us per call, with sync {(t2 - t0) / 20 / Bn * 1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:6000])
