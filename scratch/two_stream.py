"""Experiment: a batch rendered as two half-batches on two streams (list building of one half under the compositing
of the other) against one call (not a test).  usage: two_stream.py <n_img> <N> <S>"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from fresnel_amd import renderer as R
dev = torch.device('cuda:0')
n_img, N, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dist = sys.argv[4] if len(sys.argv) > 4 else "saag"
pos, scale, quat, col, opa = bench.synth_batch(n_img, N, 3000, dev, dist)
cam_t = R.pack_cameras(R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S), dev)
cfg = R._Cfg(S, S, (0.1, 0.2, 0.3), 64, False, 0.25)
gI = torch.randn(n_img, 3, S, S, device=dev); gD = torch.randn(n_img, S, S, device=dev) * 0.1

def make(lo, hi):
    return [t[lo:hi].clone().requires_grad_(True) for t in (pos, scale, quat, col, opa)]

full = make(0, n_img)
h = n_img // 2
halves = [make(0, h), make(h, n_img)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def step_full():
    for t in full: t.grad = None
    img, dep = R.GaussianRenderer.apply(*full, None, cam_t, cfg)
    torch.autograd.backward([img, dep], [gI, gD])

def step_split():
    cur = torch.cuda.current_stream()
    outs = []
    for k, (lv, st) in enumerate(zip(halves, streams)):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for t in lv: t.grad = None
            outs.append(R.GaussianRenderer.apply(*lv, None, cam_t, cfg))
    for k, (lv, st) in enumerate(zip(halves, streams)):
        with torch.cuda.stream(st):
            lo, hi = (0, h) if k == 0 else (h, n_img)
            torch.autograd.backward(list(outs[k]), [gI[lo:hi], gD[lo:hi]])
    for st in streams: cur.wait_stream(st)

def timeit(f, n=30):
    for _ in range(6): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for r in range(2):
    print(f"{n_img}x{N}@{S} {dist}: one call {timeit(step_full):.4f} ms   two half-batches on two streams {timeit(step_split):.4f} ms", flush=True)
g_full = torch.cat([t.grad.flatten() for t in full])
g_split = torch.cat([torch.cat([halves[0][i].grad, halves[1][i].grad]).flatten() for i in range(5)])
print("grads equal:", torch.equal(g_full, g_split))
