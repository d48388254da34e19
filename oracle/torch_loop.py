"""Pure-PyTorch per-Gaussian-loop restatement of the reference rasterizer -- the CPU BASELINE of
SURVEY §8(d) / BASELINE.md §4 (test infrastructure: only tests/ and bench.py's cpu_baseline leg import it).

The reference's renderer (scripts/models/differentiable_renderer.py, TileBasedRenderer.forward, DR:489-686) is a
Python loop over the depth-sorted visible Gaussians; each iteration evaluates the Gaussian on its integer bbox
(DR:594-627), clamps alpha (DR:647) and accumulates colour / depth / alpha into slices of the frame buffers
(DR:650-658); the backward is whatever autograd replays from that loop.  The reference file itself cannot travel
to the GPU box, so the timed baseline there is this restatement: the same algorithm with the same per-iteration
tensor work (a dozen small torch kernels + autograd nodes per Gaussian), written from the formulas of SURVEY §8(a)
rows a2-a9, and checked against the C oracle in tests/test_torch_loop_baseline.py.

    render_loop(...) -> image (3,H,W), depth (H,W), pairs P     (differentiable in all five inputs)
"""
import torch


def _quat_to_rot(q):
    """(N,4) wxyz, unnormalised ok -> (N,3,3)  (DR:98-120: F.normalize eps 1e-12, then the standard formula)."""
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    w, x, y, z = q.unbind(1)
    return torch.stack([
        1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y,
        2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x,
        2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y], 1).view(-1, 3, 3)


def project(pos, scale, quat, view, fx, fy, cx, cy):
    """Rows a2-a3 (DR:123-195): camera-space mean, 2-D covariance J R S S^T R^T J^T, pixel mean, depth = -z."""
    pc = pos @ view[:3, :3].T + view[:3, 3]
    x, y, z = pc.unbind(1)
    Rc = view[:3, :3] @ _quat_to_rot(quat)
    M = Rc * scale[:, None, :]
    cov3 = M @ M.transpose(1, 2)
    zs = z.abs().clamp_min(0.01) * torch.sign(z + 1e-8)
    J = torch.zeros(pos.shape[0], 2, 3, dtype=pos.dtype)
    J[:, 0, 0] = fx / (-zs)
    J[:, 0, 2] = fx * x / (zs * zs)
    J[:, 1, 1] = fy / zs
    J[:, 1, 2] = fy * y / (zs * zs)
    cov2 = J @ cov3 @ J.transpose(1, 2)
    mean2d = torch.stack([fx * x / (-zs) + cx, fy * (-y) / (-zs) + cy], 1)
    return cov2, mean2d, -z


def render_loop(pos, scale, quat, color, opacity, view, fx, fy, cx, cy, W, H, bg=(0.0, 0.0, 0.0),
                max_radius=64.0, near=0.01, far=100.0):
    """One image.  Inputs are CPU float32 tensors (requires_grad as the caller likes)."""
    cov2, mean2d, depth = project(pos, scale, quat, view, fx, fy, cx, cy)
    # radius (row a4, DR:452-487)
    a, b, c, d = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 0], cov2[:, 1, 1]
    det = (a * d - b * c).clamp_min(1e-6)
    tr = a + d
    lam = (tr + torch.sqrt((tr * tr - 4 * det).clamp_min(0))) / 2
    radius = (3 * torch.sqrt(lam.clamp_min(1e-6))).clamp_max(max_radius)
    # canonical depth order (row a5: stable, ties by index) and visibility (row a6, DR:541-562)
    order = torch.argsort(depth.detach(), stable=True)
    u, v = mean2d[:, 0].detach(), mean2d[:, 1].detach()
    r = radius.detach()
    vis = (depth.detach() > near) & (depth.detach() < far) & (u + r > 0) & (u - r < W) & (v + r > 0) & (v - r < H)
    order = order[vis[order]]
    # regularised inverse covariance, closed form (row a7; == pinv(cov + 1e-4 I) to 1.2e-6)
    ra, rd = a + 1e-4, d + 1e-4
    rdet = ra * rd - b * c
    ia, ibc, idd = rd / rdet, -(b + c) / rdet, ra / rdet
    C = torch.zeros(H, W, 3)
    A = torch.zeros(H, W)
    D = torch.zeros(H, W)
    # fp64 bbox edges with truncation, exactly like `int(mean.item() - radius.item())` (row a8, DR:594-597)
    u64, v64, r64 = u.double().tolist(), v.double().tolist(), r.double().tolist()
    pairs = 0
    for i in order.tolist():
        x0, x1 = max(0, int(u64[i] - r64[i])), min(W, int(u64[i] + r64[i]) + 1)
        y0, y1 = max(0, int(v64[i] - r64[i])), min(H, int(v64[i] + r64[i]) + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        pairs += (x1 - x0) * (y1 - y0)
        gy, gx = torch.meshgrid(torch.arange(y0, y1, dtype=torch.float32), torch.arange(x0, x1, dtype=torch.float32),
                                indexing="ij")
        dx, dy = gx - mean2d[i, 0], gy - mean2d[i, 1]
        m = ia[i] * dx * dx + ibc[i] * dx * dy + idd[i] * dy * dy
        alpha = torch.clamp(torch.exp(-0.5 * m) * opacity[i], 0, 0.99)          # row a9, DR:623-647
        w = alpha * (1.0 - A[y0:y1, x0:x1])
        C[y0:y1, x0:x1] += w.unsqueeze(-1) * color[i].view(1, 1, 3)
        D[y0:y1, x0:x1] += w * depth[i]
        A[y0:y1, x0:x1] += w
    C = C + (1.0 - A).unsqueeze(-1) * torch.tensor(bg, dtype=torch.float32).view(1, 1, 3)
    image = torch.clamp(C.permute(2, 0, 1), 0, 1)
    return image, D, pairs


def timed_fwd_bwd(arrs, view, fx, fy, cx, cy, W, H, gI, gD, threads):
    """bench.py cpu_baseline leg: one forward + autograd backward of one image with `threads` torch threads.
    arrs = numpy (pos, scale, quat, color, opacity).  Returns (pairs, seconds)."""
    import time
    old = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    try:
        ts = [torch.from_numpy(a.copy()).requires_grad_(True) for a in arrs]
        t0 = time.perf_counter()
        img, dep, pairs = render_loop(*ts, torch.from_numpy(view.copy()), fx, fy, cx, cy, W, H)
        loss = (img * torch.from_numpy(gI)).sum() + (dep * torch.from_numpy(gD)).sum()
        if loss.requires_grad:
            loss.backward()
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(old)
    return pairs, dt
