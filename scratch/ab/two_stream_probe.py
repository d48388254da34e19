"""Probe: does overlapping the latency-bound stages (projection, depth sort, list building, row sums) of one half of the batch with the
compositing kernels of the other half pay?  Two independent half-batch calls on two streams against one full-batch call.
python scratch/ab/two_stream_probe.py <workload> [chunks]"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench; bench._import_compute()
from fresnel_amd import renderer as R
wl = sys.argv[1] if len(sys.argv) > 1 else 'config2'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, S, Bn = bench.WORKLOADS[wl]
dev = torch.device('cuda:0')
full = [t.requires_grad_(True) for t in bench.synth_batch(Bn, N, 1000 * int(wl[-1]), dev)]
h = Bn // K
parts = [[t.detach()[k * h:(k + 1) * h].clone().requires_grad_(True) for t in full] for k in range(K)]
cam = R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
ren = R.TileBasedRenderer(S, S).to(dev)
gI = torch.randn(Bn, 3, S, S, device=dev); gD = torch.randn(Bn, S, S, device=dev) * 0.1
gIs = [gI[k * h:(k + 1) * h].contiguous() for k in range(K)]; gDs = [gD[k * h:(k + 1) * h].contiguous() for k in range(K)]
side = [torch.cuda.Stream() for _ in range(K - 1)]
def step_full():
    for t in full: t.grad = None
    img, dep = ren(*full, cam, return_depth=True)
    torch.autograd.backward([img, dep], [gI, gD])
def step_chunks(streams=True):
    cur = torch.cuda.current_stream()
    outs = []
    for k in range(K):
        for t in parts[k]: t.grad = None
    for k in range(K):
        if k == 0 or not streams:
            outs.append(ren(*parts[k], cam, return_depth=True))
        else:
            side[k - 1].wait_stream(cur)
            with torch.cuda.stream(side[k - 1]):
                outs.append(ren(*parts[k], cam, return_depth=True))
    if streams:
        for s in side: cur.wait_stream(s)
    torch.autograd.backward([o for pair in outs for o in pair], [g for k in range(K) for g in (gIs[k], gDs[k])])
    if streams:
        for s in side: cur.wait_stream(s)
def timeit(fn, n=200):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(3):
    print(f"{wl} B={Bn}: full batch {timeit(step_full):.4f} ms | {K} chunks, one stream {timeit(lambda: step_chunks(False)):.4f} ms | {K} chunks, {K} streams {timeit(step_chunks):.4f} ms", flush=True)
