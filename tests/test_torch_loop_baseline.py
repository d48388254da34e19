"""The pure-PyTorch per-Gaussian-loop CPU baseline (oracle/torch_loop.py, timed by bench.py's cpu_baseline leg)
computes what the C oracle computes: same pair count, image / depth <= 1e-5, autograd gradients <= 1e-4 of max."""
import numpy as np
import torch

from helpers import rel_to_max, synth_aniso, synth_saag


def _both(arrs, W, H, bg, seed):
    from oracle import fgs_oracle as orc
    from oracle import torch_loop as tl
    fx = fy = 0.8 * W
    view = np.eye(4, dtype=np.float32)
    ocam = orc.make_camera(view, fx, fy, W / 2, H / 2, W, H)
    r = orc.render(*arrs, ocam, bg=bg)
    rs = np.random.RandomState(seed)
    gI = rs.standard_normal((3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    go = orc.render_backward(r, gI, gD)
    ts = [torch.from_numpy(a.copy()).requires_grad_(True) for a in arrs]
    img, dep, pairs = tl.render_loop(*ts, torch.from_numpy(view), fx, fy, W / 2, H / 2, W, H, bg=bg)
    ((img * torch.from_numpy(gI)).sum() + (dep * torch.from_numpy(gD)).sum()).backward()
    assert pairs == r.P
    assert np.abs(img.detach().numpy() - r.image).max() <= 1e-5
    assert rel_to_max(dep.detach().numpy(), r.depth) <= 1e-5
    for t, k in zip(ts, ["positions", "scales", "rotations", "colors", "opacities"]):
        assert rel_to_max(t.grad.numpy(), go[k]) <= 1e-4, k


def test_torch_loop_matches_oracle_saag():
    _both(list(synth_saag(256, 5)), 96, 80, (0.0, 0.0, 0.0), 1)


def test_torch_loop_matches_oracle_anisotropic_clamped():
    _both(list(synth_aniso(200, 6, opacity_max=1.3)), 80, 64, (0.1, 0.2, 0.3), 2)


def test_timed_leg_returns_pairs_and_seconds():
    from oracle import torch_loop as tl
    arrs = list(synth_saag(64, 7))
    P, dt = tl.timed_fwd_bwd(arrs, np.eye(4, dtype=np.float32), 51.2, 51.2, 32, 32, 64, 64,
                             np.ones((3, 64, 64), np.float32), np.ones((64, 64), np.float32), threads=1)
    assert P > 0 and dt > 0
