"""N2 profile: the library's fused spectral / stencil losses against the stock-torch formulation on a rendered batch
(8 x 3 x 512 x 512, forward + backward).  Run plain for wall-clock numbers, or under
`rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 scratch/profile/profile_losses.py` for per-kernel
times.  Prints one JSON line with ms per call and the HBM fraction of each library kernel class (algorithmic bytes)."""
import json, sys, time
import torch
sys.path.insert(0, '.')
from fresnel_amd import losses as hip
from oracle import torch_losses as ref
dev = torch.device('cuda:0')
B, C, S = 8, 3, 512
g = torch.Generator().manual_seed(0)
r0, t0 = torch.rand(B, C, S, S, generator=g).to(dev), torch.rand(B, C, S, S, generator=g).to(dev)
d0 = (torch.rand(B, S, S, generator=g) * 2 + 0.1).to(dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def step(mod, which):
    r = r0.clone().requires_grad_(True)
    if which == 'phase':
        loss = mod.PhaseRetrievalLoss()(r, t0, d0)
    elif which == 'freq':
        loss = mod.FrequencyDomainLoss()(r, t0)
    else:
        loss = mod.wave_equation_loss(r, 0.05, pixel_spacing=1.0 / S)
    loss.backward()


out = {}
n = B * C * S * S
alg = {  # algorithmic HBM bytes, forward + backward, around the FFT (which moves 2 * 16 n bytes each way on its own)
    'phase': 24 * n + 16 * n + 32 * n + (16 + 8 + 4) * n, 'freq': 24 * n + 16 * n + 32 * n + (16 + 4) * n, 'helm': 8 * n + 8 * n}
for which in ('phase', 'freq', 'helm'):
    a, b = timed(lambda: step(hip, which)), timed(lambda: step(ref, which))
    out[which] = {'hip_ms': round(a, 4), 'torch_ms': round(b, 4), 'speedup': round(b / a, 2),
                  'algorithmic_GB': round(alg[which] / 1e9, 3)}
print(json.dumps({'workload': f'{B}x{C}x{S}x{S} fwd+bwd (includes one clone of the rendered batch)', 'losses': out}))
