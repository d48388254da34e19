"""CPU ORACLE for the angular-spectrum path (TEST INFRASTRUCTURE ONLY; see fgs_oracle.c header).

Out-of-place restatement of the reference's AngularSpectrumPropagator (DR:929-1065) and
ASMWaveFieldRenderer.forward (DR:1150-1344) in torch on the CPU; gradients come from autograd of
this restatement.  The projection/visibility/bbox stages come from the C oracle (canonical
fp32), the plane assignment, splat, FFT propagation, intensity and composition are re-stated
here line by line.  PINNED against the reference's own outputs and gradients in
tests/golden/G8_*.npz and G9_*.npz (tests/test_oracle_vs_golden.py).
"""
import numpy as np
import torch

from . import fgs_oracle as orc


def _prec(f64):
    import contextlib
    return orc.fp64() if f64 else contextlib.nullcontext()


def _zero_visible(N, H, W, bg, phases, proj, need, wavelengths, depth=False):
    """The reference's zero-visible return (DR:801-808 / DR:1207-1212): `background.view(3,1,1).expand(3,H,W) + grad_anchor`."""
    out = dict(image=np.broadcast_to(np.asarray(bg, np.float32).reshape(3, 1, 1), (3, H, W)).copy(), proj=proj)
    if depth:
        out["depth"] = np.zeros((H, W), np.float32)
    if need:
        z = lambda *sh: np.zeros(sh, np.float32)
        out.update(grad_positions=z(N, 3), grad_scales=z(N, 3), grad_rotations=z(N, 4), grad_colors=z(N, 3),
                   grad_opacities=z(N), grad_phases=np.zeros(np.asarray(phases).shape, np.float32))
        if wavelengths:
            out["grad_wavelengths"] = z(3)
    return out


def _project_bwd(pos, scale, quat, cam, proj, mean, conic, dep, f64):
    """Chain dL/d(mean2d, conic, depth) (autograd of the restatement) through the C oracle's projection adjoint."""
    import ctypes
    with _prec(f64):
        real = orc._REAL
        N = len(proj["visible"])
        g = lambda x: np.zeros(tuple(x.shape), real) if (x is None or x.grad is None) else np.ascontiguousarray(x.grad.numpy(), dtype=real)
        gm, gc = g(mean), g(conic)
        gd = np.zeros(N, real) if dep is None else g(dep)
        g_pos, g_scale, g_quat = np.zeros((N, 3), real), np.zeros((N, 3), real), np.zeros((N, 4), real)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        p_, s_, q_ = (np.ascontiguousarray(a, dtype=real) for a in (pos, scale, quat))
        orc.lib().fgs_or_project_bwd(ctypes.c_int32(N), P(p_), P(s_), P(q_), orc._camref(cam),
                                     P(proj["visible"]), P(gm), P(gc), P(gd), P(g_pos), P(g_scale), P(g_quat))
    return g_pos.astype(np.float32), g_scale.astype(np.float32), g_quat.astype(np.float32)


def transfer_function(H, W, pixel_pitch, z, wl, dtype=torch.float32):
    """DR:959-961 + DR:989-999."""
    fx = torch.fft.fftfreq(W, d=pixel_pitch, dtype=dtype)  # (the propagator's buffers take the default dtype of the run, DR:959-961)
    fy = torch.fft.fftfreq(H, d=pixel_pitch, dtype=dtype)
    FX, FY = torch.meshgrid(fx, fy, indexing="xy")
    kz_sq = torch.clamp((1.0 / wl) ** 2 - FX ** 2 - FY ** 2, min=0)
    return torch.exp(1j * 2 * torch.pi * z * torch.sqrt(kz_sq))


def propagate(field, z, wl, pixel_pitch=1.0 / 256.0):
    """DR:1041-1047 for one (H,W) complex field."""
    H, W = field.shape
    rdt = torch.float64 if field.dtype == torch.complex128 else torch.float32
    Htf = transfer_function(H, W, pixel_pitch, torch.as_tensor(z, dtype=rdt), torch.as_tensor(wl, dtype=rdt), rdt)
    return torch.fft.ifft2(torch.fft.fft2(field) * Htf)


def render(pos, scale, quat, color, opacity, phases, wavelengths, cam, bg=(0.0, 0.0, 0.0), max_radius=64.0,
           num_planes=16, depth_range=(0.1, 2.0), focal_depth=0.5, pixel_pitch=1.0 / 256.0,
           dtype=torch.float32, grad_out=None, project_f64=False):
    """ASM forward for one image; with grad_out (3,H,W) also returns gradients of sum(img*grad_out)
    w.r.t. mean2d/conic/opacity/colour/phase/wavelengths chained through the C oracle's projection
    backward to positions/scales/rotations."""
    with _prec(project_f64):  # (project_f64: the projection and its adjoint on the fp64 referee build too -- the sweeps' referee)
        proj = orc.project(pos, scale, quat, cam, max_radius)
    W, H = cam.width, cam.height
    vis = proj["visible"].astype(bool)
    N = len(vis)
    if not vis.any():
        # DR:1207-1212: no visible Gaussian -> the background itself (NOT background + sqrt(1e-8): the intensity floor of DR:1319
        # is never reached), every gradient zero.  (Missing until round 5: the sweeps' two N = 1 cases with the one Gaussian
        # culled showed exactly 1.00e-4 between this oracle and the HIP path, which follows the reference.)
        return _zero_visible(N, H, W, bg, phases, proj, grad_out is not None, wavelengths=True)
    t = lambda a, g=False: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=g)
    need = grad_out is not None
    mean, conic = t(proj["mean2d"], need), t(proj["conic"], need)
    opa, col, ph, wl = t(opacity, need), t(color, need), t(phases, need), t(wavelengths, need)
    # DR:1106: the plane depths are a buffer of the module's default dtype -- fp32 in the reference's fp32 run, fp64 in its fp64 run
    # (round 5: until then this restatement kept them, and the propagation distances below, in fp32 in BOTH modes, and its "fp64" run
    # sat 1.7e-5 ... 2e-4 (dL/dlambda) from the reference's own fp64 run on K6: z rounded to fp32 under a phase of ~200 rad)
    planes = torch.linspace(depth_range[0], depth_range[1], num_planes, dtype=dtype)
    depth_t = torch.tensor(proj["depth"], dtype=dtype)
    plane_idx = (depth_t.unsqueeze(1) - planes.unsqueeze(0)).abs().argmin(dim=1)  # DR:1147-1148
    fields = [[torch.zeros(H, W, dtype=dtype), torch.zeros(H, W, dtype=dtype)] for _ in range(num_planes * 3)]
    for i in range(N):  # DR:1238-1283 (order-independent accumulation)
        if not vis[i]:
            continue
        x0, x1, y0, y1 = [int(v) for v in proj["bbox"][i]]
        if x0 >= x1 or y0 >= y1:
            continue
        ly, lx = torch.meshgrid(torch.arange(y0, y1, dtype=dtype), torch.arange(x0, x1, dtype=dtype), indexing="ij")
        dx, dy = lx - mean[i, 0], ly - mean[i, 1]
        m = conic[i, 0] * dx * dx + conic[i, 1] * dx * dy + conic[i, 2] * dy * dy
        amp = torch.exp(-0.5 * m) * opa[i]
        p = int(plane_idx[i])
        for c in range(3):
            phc = ph[i, c] if ph.dim() == 2 else ph[i]
            pad = (x0, W - x1, y0, H - y1)
            fields[p * 3 + c][0] = fields[p * 3 + c][0] + torch.nn.functional.pad(amp * col[i, c] * torch.cos(phc), pad)
            fields[p * 3 + c][1] = fields[p * 3 + c][1] + torch.nn.functional.pad(amp * col[i, c] * torch.sin(phc), pad)
    total = [torch.zeros(H, W, dtype=torch.complex128 if dtype == torch.float64 else torch.complex64) for _ in range(3)]
    for p in range(num_planes):  # DR:1291-1313
        z = torch.tensor(focal_depth, dtype=dtype) - planes[p]                      # DR:1293-1295, in the run's dtype
        fc = [torch.complex(fields[p * 3 + c][0], fields[p * 3 + c][1]) for c in range(3)]
        if max(float(f.detach().abs().max()) for f in fc) < 1e-8:
            continue
        for c in range(3):
            Htf = transfer_function(H, W, pixel_pitch, z.to(dtype), wl[c], dtype)
            total[c] = total[c] + torch.fft.ifft2(torch.fft.fft2(fc[c]) * Htf)
    tf = torch.stack(total, dim=-1)                                   # (H,W,3) complex
    rendered = torch.sqrt(tf.real ** 2 + tf.imag ** 2 + 1e-8)          # DR:1316-1319
    rendered = torch.clamp(rendered / rendered.max().clamp(min=1.0), 0, 1)
    total_amp = tf.abs().sum(dim=-1, keepdim=True).clamp(0, 1)         # DR:1327
    rendered = rendered + torch.tensor(bg, dtype=dtype).view(1, 1, 3) * (1 - total_amp)
    img = torch.clamp(rendered.permute(2, 0, 1), 0, 1)
    out = dict(image=img.detach().float().numpy(), plane_idx=plane_idx.numpy(), proj=proj)
    if need:
        loss = (img * torch.tensor(grad_out, dtype=dtype)).sum()
        if loss.requires_grad:  # (no visible Gaussian: the image is a constant, every gradient is zero)
            loss.backward()
        z = lambda x: np.zeros_like(np.asarray(x.detach()), dtype=np.float32) if x.grad is None else x.grad.float().numpy()
        g_pos, g_scale, g_quat = _project_bwd(pos, scale, quat, cam, proj, mean, conic, None, project_f64)
        out.update(grad_positions=g_pos, grad_scales=g_scale, grad_rotations=g_quat, grad_colors=z(col),
                   grad_opacities=z(opa), grad_phases=z(ph), grad_wavelengths=z(wl))
    return out


def render_wave(pos, scale, quat, color, opacity, phases, cam, bg=(0.0, 0.0, 0.0), max_radius=64.0,
                dtype=torch.float32, grad_out=None, grad_depth=None, project_f64=False):
    """WaveFieldRenderer (DR:747-926) for one image: order-independent complex accumulation, intensity,
    max normalisation, background, amplitude-weighted depth map.  Gradients by autograd of this
    restatement, chained through the C oracle's projection backward."""
    with _prec(project_f64):  # (project_f64: the projection and its adjoint on the fp64 referee build too -- the sweeps' referee)
        proj = orc.project(pos, scale, quat, cam, max_radius)
    W, H = cam.width, cam.height
    vis = proj["visible"].astype(bool)
    N = len(vis)
    need = grad_out is not None
    if not vis.any():  # DR:801-808: background, zero depth map, zero gradients
        return _zero_visible(N, H, W, bg, phases, proj, need, wavelengths=False, depth=True)
    t = lambda a, g=False: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=g)
    mean, conic, dep = t(proj["mean2d"], need), t(proj["conic"], need), t(proj["depth"], need)
    opa, col, ph = t(opacity, need), t(color, need), t(phases, need)
    re = [torch.zeros(H, W, dtype=dtype) for _ in range(3)]
    im = [torch.zeros(H, W, dtype=dtype) for _ in range(3)]
    ad, wt = torch.zeros(H, W, dtype=dtype), torch.zeros(H, W, dtype=dtype)
    for i in range(N):
        if not vis[i]:
            continue
        x0, x1, y0, y1 = [int(v) for v in proj["bbox"][i]]
        if x0 >= x1 or y0 >= y1:
            continue
        ly, lx = torch.meshgrid(torch.arange(y0, y1, dtype=dtype), torch.arange(x0, x1, dtype=dtype), indexing="ij")
        dx, dy = lx - mean[i, 0], ly - mean[i, 1]
        m = conic[i, 0] * dx * dx + conic[i, 1] * dx * dy + conic[i, 2] * dy * dy
        amp = torch.exp(-0.5 * m) * opa[i]
        pad = (x0, W - x1, y0, H - y1)
        for c in range(3):
            phc = ph[i, c] if ph.dim() == 2 else ph[i]
            re[c] = re[c] + torch.nn.functional.pad(amp * col[i, c] * torch.cos(phc), pad)
            im[c] = im[c] + torch.nn.functional.pad(amp * col[i, c] * torch.sin(phc), pad)
        ad = ad + torch.nn.functional.pad(amp * dep[i], pad)
        wt = wt + torch.nn.functional.pad(amp, pad)
    wr, wi = torch.stack(re, -1), torch.stack(im, -1)
    rendered = torch.sqrt(wr ** 2 + wi ** 2 + 1e-8)
    rendered = torch.clamp(rendered / rendered.max().clamp(min=1.0), 0, 1)
    ta = torch.sqrt((wr ** 2 + wi ** 2).sum(dim=-1, keepdim=True) + 1e-8).clamp(0, 1)
    rendered = rendered + torch.tensor(bg, dtype=dtype).view(1, 1, 3) * (1 - ta)
    img = torch.clamp(rendered.permute(2, 0, 1), 0, 1)
    dmap = ad / (wt + 1e-8)
    out = dict(image=img.detach().float().numpy(), depth=dmap.detach().float().numpy(), proj=proj)
    if need:
        loss = (img * torch.tensor(grad_out, dtype=dtype)).sum()
        if grad_depth is not None:
            loss = loss + (dmap * torch.tensor(grad_depth, dtype=dtype)).sum()
        if loss.requires_grad:  # (no visible Gaussian: the image is a constant, every gradient is zero)
            loss.backward()
        z = lambda x: np.zeros_like(np.asarray(x.detach()), dtype=np.float32) if x.grad is None else x.grad.float().numpy()
        g_pos, g_scale, g_quat = _project_bwd(pos, scale, quat, cam, proj, mean, conic, dep, project_f64)
        out.update(grad_positions=g_pos, grad_scales=g_scale, grad_rotations=g_quat, grad_colors=z(col),
                   grad_opacities=z(opa), grad_phases=z(ph))
    return out
