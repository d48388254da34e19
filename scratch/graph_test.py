"""Does capturing forward + backward of the rasterizer in a HIP graph (torch.cuda.CUDAGraph) pay?  The library never
allocates or synchronises, so the whole step is capturable; this measures replay vs eager for a workload."""
import sys, time, torch
sys.path.insert(0, '.')
import bench as Bn
from fresnel_amd import renderer as R
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'config2'
N, S, B = Bn.WORKLOADS[name]
pos, scale, quat, col, opa = Bn.synth_batch(B, N, 1000, dev)
leaves = [t.requires_grad_(True) for t in (pos, scale, quat, col, opa)]
cam = R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
ren = R.TileBasedRenderer(S, S).to(dev)
gI = torch.randn(B, 3, S, S, device=dev); gD = torch.randn(B, S, S, device=dev) * 0.1
grads = [torch.zeros_like(t) for t in leaves]


def step():
    img, dep = ren(*leaves, cam, return_depth=True)
    gs = torch.autograd.grad([img, dep], leaves, [gI, gD])
    for g, o in zip(gs, grads):
        o.copy_(g)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


eager = timed(step)
ref = [g.clone() for g in grads]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
graphed = timed(g.replay)
ok = all(torch.equal(a, b) for a, b in zip(ref, grads))
print(f"{name}: eager {eager:.4f} ms  graph replay {graphed:.4f} ms  identical gradients {ok}")
