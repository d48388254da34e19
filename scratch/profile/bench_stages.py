import sys, os, time, json, numpy as np, torch
sys.path.insert(0, '.')
from fresnel_amd import _binding as B, renderer as R
import bench; bench._import_compute()
dev = torch.device('cuda:0')
def run(N, S, nimg, steps=10, warm=3, dist='saag'):
    pos, scale, quat, col, opa = bench.synth_batch(nimg, N, 3000, dev)
    if dist == 'decoder':   # "decoder-like" distribution of SURVEY 8d
        s = int(np.sqrt(N)); N2 = s * s
        g = torch.Generator().manual_seed(7)
        xs = torch.linspace(-1, 1, s)
        gx, gy = torch.meshgrid(xs, xs, indexing='xy')
        pos = torch.stack([gx.reshape(-1), gy.reshape(-1), torch.zeros(N2)], -1)[None].repeat(nimg, 1, 1)
        pos[..., 2] = -2 - 2 * torch.rand(nimg, N2, generator=g)
        scale = 0.13 + 0.03 * torch.rand(nimg, N2, 3, generator=g)
        quat = torch.randn(nimg, N2, 4, generator=g)
        col = torch.rand(nimg, N2, 3, generator=g); opa = torch.rand(nimg, N2, generator=g)
        pos, scale, quat, col, opa = [t.to(dev).contiguous() for t in (pos, scale, quat, col, opa)]
    leaves = [t.requires_grad_(True) for t in (pos, scale, quat, col, opa)]
    cam_t = R.pack_cameras(R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S), dev)
    cfg = R._Cfg(S, S, (0, 0, 0), 64, False, 0.25)
    gI = torch.randn(nimg, 3, S, S, device=dev); gD = torch.randn(nimg, S, S, device=dev) * 0.1
    def step():
        for t in leaves: t.grad = None
        img, dep = R.GaussianRenderer.apply(*leaves, None, cam_t, cfg)
        torch.autograd.backward([img, dep], [gI, gD])
    for _ in range(warm): step()
    torch.cuda.synchronize()
    _, _, saved, dims, _ = R.forward_raw(*[t.detach() for t in leaves], None, cam_t, cfg)
    st = R.inspect_saved(saved, dims); D = int(st['counters'][0].item())
    rg = st['ranges'].cpu().numpy().astype(np.int64); cnt = (rg[..., 1] - rg[..., 0]).reshape(-1)
    import ctypes
    pairs = torch.zeros(1, dtype=torch.int64, device=dev)
    B.check(B.load().fgs_count_pairs(ctypes.byref(dims), ctypes.c_void_p(saved.data_ptr()), ctypes.c_void_p(pairs.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'cp')
    P = int(pairs.item()); del saved, st
    B.stage_timing_enable(True); B.stage_timing_read()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps * 1e3
    stg = B.stage_timing_read(); B.stage_timing_enable(False)
    avg = {k: round(v[0] / max(v[1], 1), 4) for k, v in stg.items()}
    print(json.dumps(dict(N=N, S=S, B=nimg, dist=dist, ms=round(el, 3), pairs=P, D=D, Gpairs_s=round(P / el / 1e6, 1),
                          maxlist=int(cnt.max()), meanlist=float(cnt.mean()), stages=avg)))
if __name__ == '__main__':
    for args in [(32768, 512, 8), (32768, 512, 16), (32768, 512, 32), (8192, 256, 16), (32761, 512, 8, 10, 3, 'decoder')]:
        run(*args)
