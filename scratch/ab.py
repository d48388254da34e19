import sys, os, json, numpy as np, torch
sys.path.insert(0, '.')
from fresnel_amd import _binding as B, renderer as R
import bench
dev = torch.device('cuda:0')
def setup(N, S, nimg):
    pos, scale, quat, col, opa = bench.synth_batch(nimg, N, 3000, dev)
    leaves = [t.requires_grad_(True) for t in (pos, scale, quat, col, opa)]
    cam_t = R.pack_cameras(R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S), dev)
    cfg = R._Cfg(S, S, (0, 0, 0), 64, False, 0.25)
    gI = torch.randn(nimg, 3, S, S, device=dev); gD = torch.randn(nimg, S, S, device=dev) * 0.1
    def step():
        for t in leaves: t.grad = None
        img, dep = R.GaussianRenderer.apply(*leaves, None, cam_t, cfg)
        torch.autograd.backward([img, dep], [gI, gD])
    return step
def ab(step, envname, variants, stage, rounds=7, steps=4):
    res = {v: [] for v in variants}
    B.stage_timing_enable(True)
    for v in variants:
        os.environ[envname] = str(v); step()
    torch.cuda.synchronize(); B.stage_timing_read()
    for r in range(rounds):
        for v in variants:
            os.environ[envname] = str(v)
            for _ in range(steps): step()
            torch.cuda.synchronize()
            st = B.stage_timing_read()
            res[v].append(st[stage][0] / max(st[stage][1], 1))
    B.stage_timing_enable(False)
    return {v: (round(float(np.median(x)), 4), round(float(np.min(x)), 4)) for v, x in res.items()}
if __name__ == '__main__':
    for nimg in (8, 32):
        step = setup(32768, 512, nimg)
        print('B', nimg, 'fwd variants (median,min ms):', ab(step, 'FGS_FWD_WAVES', [1, 2], 'composite_fwd'))
        os.environ['FGS_FWD_WAVES'] = '2'
        pass
        os.environ['FGS_BWD_VARIANT'] = '3'
