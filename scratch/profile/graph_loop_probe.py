"""Route A under a HIP graph: the reference's per-image loop (B sequential (N,.) module calls + torch.stack + backward, TGD:1209-1226)
captured ONCE with torch.cuda.CUDAGraph and replayed -- what a maintainer who keeps the loop can do about its host cost.
python scratch/profile/graph_loop_probe.py [workload]"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench; bench._import_compute()
from fresnel_amd import renderer as R
wl = sys.argv[1] if len(sys.argv) > 1 else 'config1'
N, S, Bn = bench.WORKLOADS[wl]
dev = torch.device('cuda:0')
leaves = [t.requires_grad_(True) for t in bench.synth_batch(Bn, N, 1000, dev)]
cam = R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
ren = R.TileBasedRenderer(S, S).to(dev)
gI = torch.randn(Bn, 3, S, S, device=dev); gD = torch.randn(Bn, S, S, device=dev) * 0.1
def step():
    imgs, deps = [], []
    for b in range(Bn):
        im, dp = ren(*[t[b] for t in leaves], cam, return_depth=True)
        imgs.append(im); deps.append(dp)
    torch.autograd.backward([torch.stack(imgs), torch.stack(deps)], [gI, gD])
def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for t in leaves: t.grad = torch.zeros_like(t)   # static gradient buffers: the graph accumulates into them
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
eager = timeit(step, 30)
for t in leaves: t.grad.zero_()
step(); torch.cuda.synchronize()
ref = [t.grad.clone() for t in leaves]            # one eager step's gradients
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
for t in leaves: t.grad.zero_()
g.replay(); torch.cuda.synchronize()
same = all(torch.equal(a, t.grad) for a, t in zip(ref, leaves))
def replay():
    g.replay()
graphed = timeit(replay, 100)
print(f"{wl}: per-image loop of {Bn} images, eager {eager:.3f} ms per step | captured in one HIP graph and replayed {graphed:.3f} ms per step | "
      f"gradients of one replay bit-equal to one eager step's: {same}")
