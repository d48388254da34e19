#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the reference's Python renderer from /root/reference/scripts (read-only),
runs it on small seeded inputs and writes inputs + reference outputs as .npz
fixtures next to this file.  Nothing from the reference is copied: the fixtures
hold data only (inputs, expected outputs, gradients, generating-stack metadata).

Cases:
  G1 dummy-SAAG N=256 @128x128                (config-1 shape)
  G2 anisotropic, opacity up to 1.3 (0.99 clamp active), bg != 0, N=300 @96x96
  G3 all Gaussians behind the camera -> background, zero grads
  G4 huge scales -> radius cap 64 active, @160x160
  G5 zone-quantised depths (8 values), argsort patched to stable=True in THIS harness
  G6 G1 + scalar phases, use_phase_blending=True (reference forward + colour-only grad;
     full grads from an out-of-place restatement that reproduces the forward exactly)
  G7 off-axis view matrix (orbit camera el=20deg az=135deg)
  G8 AngularSpectrumPropagator 64x64 known answers
  G9 ASMWaveFieldRenderer N=256 @128x128, scalar and (N,3) phases, grads incl. wavelengths
  G10 WaveFieldRenderer N=256 @128x128, scalar and (N,3) phases, image + depth + grads
  G11 FFT / stencil losses on a rendered batch: PhaseRetrievalLoss, FrequencyDomainLoss,
      wave_equation_loss -- values and input gradients on (2,3,48,40) batches
  G12 caller-side hand-off pieces (HFTSConfig, orbit cameras, ImageDataset)
  G13 mid-size scene: 1024 Gaussians @256x256, radius cap active, several hundred entries per pixel (tile lists of
      >= 4 depth segments): the long accumulation chains are pinned by the reference, not only N <= 400
  G15 ONE image of the headline configuration (BASELINE config 3: 32 768 Gaussians @512x512, create_dummy_saag distribution)
      through the reference itself: image / depth rows 0::16 and the gradients of every 8th Gaussian; inputs are regenerated
      from the seed by tests/helpers.synth_saag
  G16 ONE image of BASELINE config 5 AS BENCHMARKED (8 192 Gaussians @512x512, 16 depth planes, per-channel wavelengths, scalar
      phases) through ASMWaveFieldRenderer in fp32 and fp64 (16 + 29 minutes here): image rows 0::16, gradients of every 8th
      Gaussian, dL/dlambda; G9f64: the fp64 run of the two G9 scenes (referee of dL/dlambda, which is NaN in fp32 for lambda = .05)
  K1-K5 the randomized sweeps' known kink / conditioning cases (tests/fuzz_cases.py replays the sweep's draws): two
      phase-path scenes and two ASM scenes, each through the reference-derived referee in fp32 and in fp64
  G14 needles / discs at scale ratios 30:1, 100:1, 500:1 through the reference in fp32 AND in fp64 (default dtype
      switched to float64 in this harness): which gradients are well-conditioned enough for a 1e-4 statement

Upstream gradients gI ~ N(0,1), gD ~ N(0,0.01) come from numpy's frozen legacy
RandomState(seed) so tests can regenerate them bit-exactly; they are stored too.
"""
import os
import sys

REF = "/root/reference/scripts"
if not os.path.isdir(REF):
    sys.exit("make_goldens.py: /root/reference is absent; goldens can only be generated "
             "in the build container")
sys.path.insert(0, REF)

import numpy as np
import torch

from models.differentiable_renderer import (  # noqa: E402  (the reference, read-only)
    Camera, TileBasedRenderer, compute_2d_covariance,
    AngularSpectrumPropagator, ASMWaveFieldRenderer, WaveFieldRenderer,
)

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)
META = dict(torch=torch.__version__, numpy=np.__version__, device="cpu")


def upstream(seed, H, W):
    rs = np.random.RandomState(seed)
    gI = rs.standard_normal((3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)  # var 0.01
    return gI, gD


def frontal_camera(R):
    return Camera(fx=0.8 * R, fy=0.8 * R, cx=R / 2, cy=R / 2, width=R, height=R)


def orbit_view(el_deg, az_deg, distance=2.0):
    """View matrix of the reference's orbit camera (formula at
    scripts/training/train_gaussian_decoder.py:706-744), evaluated here in numpy."""
    el, az = np.deg2rad(el_deg), np.deg2rad(az_deg)
    cam = np.array([distance * np.cos(el) * np.sin(az), distance * np.sin(el),
                    distance * np.cos(el) * np.cos(az)])
    fwd = -cam / np.linalg.norm(cam)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    Rm = np.array([right, up, -fwd])
    t = -Rm @ cam
    V = np.eye(4, dtype=np.float32)
    V[:3, :3] = Rm.astype(np.float32)
    V[:3, 3] = t.astype(np.float32)
    return V


def dummy_saag(N, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.randn(N, 3, generator=g) * 0.5
    pos[:, 2] -= 2
    scale = torch.ones(N, 3) * 0.05
    rot = torch.zeros(N, 4)
    rot[:, 0] = 1
    col = torch.rand(N, 3, generator=g)
    opa = torch.ones(N) * 0.8
    return pos, scale, rot, col, opa


def intermediates(renderer, pos, scale, rot, camera, stable):
    """Re-evaluate the reference's integer stages with its own helper functions."""
    with torch.no_grad():
        cov, mean, depth = compute_2d_covariance(pos, scale, rot, camera)
        radii = renderer._compute_radius(cov)
        order = torch.argsort(depth, stable=stable)
        order_unstable = torch.argsort(depth)
        W, H = renderer.width, renderer.height
        vis = (depth > camera.near) & (depth < camera.far)
        vis &= (mean[:, 0] + radii > 0) & (mean[:, 0] - radii < W)
        vis &= (mean[:, 1] + radii > 0) & (mean[:, 1] - radii < H)
        bbox = np.zeros((pos.shape[0], 4), dtype=np.int32)
        for i in range(pos.shape[0]):
            if not bool(vis[i]):
                continue
            r = radii[i].item()
            x0 = max(0, int(mean[i, 0].item() - r))
            x1 = min(W, int(mean[i, 0].item() + r) + 1)
            y0 = max(0, int(mean[i, 1].item() - r))
            y1 = min(H, int(mean[i, 1].item() + r) + 1)
            bbox[i] = (x0, x1, y0, y1)
        cov_reg = cov + 1e-4 * torch.eye(2).unsqueeze(0)
        inv = torch.linalg.pinv(cov_reg)
    return dict(cov_2d=cov.numpy(), means_2d=mean.numpy(), depths=depth.numpy(),
                radii=radii.numpy(), depth_order=order.numpy().astype(np.int32),
                depth_order_unstable=order_unstable.numpy().astype(np.int32),
                visible=vis.numpy().astype(np.uint8), bbox=bbox, cov_inv=inv.numpy())


def run_tbr(name, pos, scale, rot, col, opa, camera, H, W, bg=(0.0, 0.0, 0.0),
            seed_up=0, phases=None, use_phase=False, amp=0.25, stable=False,
            grad_inputs="all"):
    renderer = TileBasedRenderer(W, H, background=bg, use_phase_blending=use_phase,
                                 phase_amplitude=amp)
    leaves = [t.clone().requires_grad_(grad_inputs == "all") for t in (pos, scale, rot, col, opa)]
    if grad_inputs == "color":
        leaves[3].requires_grad_(True)
    ph = None
    if phases is not None:
        ph = phases.clone()
    real_argsort = torch.argsort
    if stable:
        torch.argsort = lambda x, *a, **k: real_argsort(x, stable=True)
    try:
        img, dep = renderer(*leaves, camera, return_depth=True, phases=ph)
    finally:
        torch.argsort = real_argsort
    gI, gD = upstream(seed_up, H, W)
    loss = (img * torch.from_numpy(gI)).sum() + (dep * torch.from_numpy(gD)).sum()
    loss.backward()
    rec = dict(
        name=name, positions=pos.numpy(), scales=scale.numpy(), rotations=rot.numpy(),
        colors=col.numpy(), opacities=opa.numpy(),
        view=camera.view_matrix.numpy().astype(np.float32),
        intr=np.array([camera.fx, camera.fy, camera.cx, camera.cy, camera.near, camera.far],
                      dtype=np.float64),
        size=np.array([W, H], dtype=np.int32), background=np.array(bg, dtype=np.float32),
        seed_up=np.int32(seed_up), gI=gI, gD=gD,
        image=img.detach().numpy(), depth=dep.detach().numpy(),
        use_phase=np.uint8(use_phase), phase_amplitude=np.float32(amp), stable=np.uint8(stable),
    )
    names = ["positions", "scales", "rotations", "colors", "opacities"]
    for n, t in zip(names, leaves):
        if t.grad is not None:
            rec["grad_" + n] = t.grad.numpy()
        elif t.requires_grad:
            rec["grad_" + n] = np.zeros_like(t.detach().numpy())
    if phases is not None:
        rec["phases"] = phases.numpy()
    rec.update(intermediates(renderer, pos, scale, rot, camera, stable))
    for k, v in META.items():
        rec["meta_" + k] = np.array(v)
    return rec


def save(rec, fname):
    path = os.path.join(OUT, fname)
    np.savez_compressed(path, **rec)
    print(f"{fname}: {os.path.getsize(path) / 1024:.0f} KB")


# ----------------------------------------------------------------------------------
# G6 helper: an out-of-place restatement of the phase-blending recurrence (our code,
# written for this harness) whose autograd supplies the full gradients the reference
# cannot produce.  It must reproduce the reference forward exactly.
# ----------------------------------------------------------------------------------
def phase_restatement(pos, scale, rot, col, opa, phases, camera, H, W, bg, amp, inter):
    cov, mean, depth = compute_2d_covariance(pos, scale, rot, camera)
    order = torch.from_numpy(inter["depth_order"].astype(np.int64))
    vis = torch.from_numpy(inter["visible"].astype(bool))
    bbox = inter["bbox"]
    cov_reg = cov + 1e-4 * torch.eye(2).unsqueeze(0)
    inv = torch.linalg.pinv(cov_reg)
    C = torch.zeros(H, W, 3)
    A = torch.zeros(H, W)
    D = torch.zeros(H, W)
    P = torch.zeros(H, W)
    for i in order.tolist():
        if not bool(vis[i]):
            continue
        x0, x1, y0, y1 = [int(t) for t in bbox[i]]
        if x0 >= x1 or y0 >= y1:
            continue
        ly, lx = torch.meshgrid(torch.arange(y0, y1, dtype=torch.float32),
                                torch.arange(x0, x1, dtype=torch.float32), indexing="ij")
        dx = lx - mean[i, 0]
        dy = ly - mean[i, 1]
        a, b, c, d = inv[i, 0, 0], inv[i, 0, 1], inv[i, 1, 0], inv[i, 1, 1]
        mahal = a * dx * dx + (b + c) * dx * dy + d * dy * dy
        alpha = torch.exp(-0.5 * mahal) * opa[i]
        prev = P[y0:y1, x0:x1]
        pd = torch.abs(phases[i] - prev)
        pd = torch.min(pd, 1.0 - pd)
        alpha = alpha * ((1.0 - amp) + amp * torch.cos(pd * 2 * 3.14159))
        alpha = torch.clamp(alpha, 0, 0.99)
        T = 1.0 - A[y0:y1, x0:x1]
        w = alpha * T
        mask = torch.zeros(H, W, dtype=torch.bool)
        mask[y0:y1, x0:x1] = True
        wfull = torch.zeros(H, W).masked_scatter(mask, w)
        C = C + wfull.unsqueeze(-1) * col[i].view(1, 1, 3)
        D = D + wfull * depth[i]
        A = A + wfull
        pc = wfull / A.clamp(min=1e-6)
        Pn = P * (1 - pc) + phases[i] * pc
        P = torch.where(mask, Pn, P)
    C = C + (1.0 - A).unsqueeze(-1) * torch.tensor(bg).view(1, 1, 3)
    return torch.clamp(C.permute(2, 0, 1), 0, 1), D


def main():
    # G1 ---------------------------------------------------------------------------
    R = 128
    pos, scale, rot, col, opa = dummy_saag(256, 0)
    g1 = run_tbr("G1", pos, scale, rot, col, opa, frontal_camera(R), R, R, seed_up=101)
    save(g1, "G1_saag256_128.npz")

    # G2 ---------------------------------------------------------------------------
    R = 96
    g = torch.Generator().manual_seed(2)
    N = 300
    pos2 = torch.randn(N, 3, generator=g) * 0.5
    pos2[:, 2] -= 2
    scale2 = torch.rand(N, 3, generator=g) * 0.12 + 0.01
    rot2 = torch.randn(N, 4, generator=g)
    col2 = torch.rand(N, 3, generator=g)
    opa2 = torch.rand(N, generator=g) * 1.3
    g2 = run_tbr("G2", pos2, scale2, rot2, col2, opa2, frontal_camera(R), R, R,
                 bg=(0.1, 0.2, 0.3), seed_up=102)
    save(g2, "G2_aniso300_96.npz")

    # G3 ---------------------------------------------------------------------------
    R = 64
    pos3, scale3, rot3, col3, opa3 = dummy_saag(64, 3)
    pos3 = pos3.clone()
    pos3[:, 2] += 4  # z ~ +2: behind the camera (camera looks down -Z)
    g3 = run_tbr("G3", pos3, scale3, rot3, col3, opa3, frontal_camera(R), R, R,
                 bg=(0.25, 0.5, 0.75), seed_up=103)
    save(g3, "G3_behind64_64.npz")

    # G4 ---------------------------------------------------------------------------
    R = 160
    g = torch.Generator().manual_seed(4)
    N = 96
    pos4 = torch.randn(N, 3, generator=g) * 0.6
    pos4[:, 2] -= 2
    scale4 = torch.rand(N, 3, generator=g) * 0.6 + 0.2
    rot4 = torch.randn(N, 4, generator=g)
    col4 = torch.rand(N, 3, generator=g)
    opa4 = torch.rand(N, generator=g) * 0.5
    g4 = run_tbr("G4", pos4, scale4, rot4, col4, opa4, frontal_camera(R), R, R, seed_up=104)
    save(g4, "G4_radcap96_160.npz")

    # G5 ---------------------------------------------------------------------------
    R = 96
    g = torch.Generator().manual_seed(5)
    N = 400
    pos5 = torch.randn(N, 3, generator=g) * 0.5
    zone = torch.randint(0, 8, (N,), generator=g).float()
    pos5[:, 2] = -2.0 - 2.0 * (zone + 0.5) / 8.0
    scale5 = torch.rand(N, 3, generator=g) * 0.06 + 0.03
    rot5 = torch.randn(N, 4, generator=g)
    col5 = torch.rand(N, 3, generator=g)
    opa5 = torch.rand(N, generator=g) * 0.9 + 0.05
    g5 = run_tbr("G5", pos5, scale5, rot5, col5, opa5, frontal_camera(R), R, R,
                 seed_up=105, stable=True)
    save(g5, "G5_zones400_96.npz")

    # G6 ---------------------------------------------------------------------------
    R = 128
    pos6, scale6, rot6, col6, opa6 = dummy_saag(256, 0)
    g = torch.Generator().manual_seed(6)
    ph6 = torch.rand(256, generator=g)
    cam6 = frontal_camera(R)
    g6 = run_tbr("G6", pos6, scale6, rot6, col6, opa6, cam6, R, R, seed_up=106,
                 phases=ph6, use_phase=True, amp=0.25, grad_inputs="color")
    leaves = [t.clone().requires_grad_(True) for t in (pos6, scale6, rot6, col6, opa6, ph6)]
    img_r, dep_r = phase_restatement(*leaves, cam6, R, R, (0.0, 0.0, 0.0), 0.25, g6)
    fwd_diff = max(float((img_r.detach() - torch.from_numpy(g6["image"])).abs().max()),
                   float((dep_r.detach() - torch.from_numpy(g6["depth"])).abs().max()))
    print("G6 restatement vs reference forward: max abs diff =", fwd_diff)
    assert fwd_diff == 0.0, "out-of-place restatement must reproduce the reference forward"
    loss = (img_r * torch.from_numpy(g6["gI"])).sum() + (dep_r * torch.from_numpy(g6["gD"])).sum()
    loss.backward()
    col_diff = float((leaves[3].grad - torch.from_numpy(g6["grad_colors"])).abs().max())
    print("G6 restatement colour-grad vs reference colour-grad: max abs diff =", col_diff)
    g6["ref_grad_colors"] = g6.pop("grad_colors")
    for n, t in zip(["positions", "scales", "rotations", "colors", "opacities", "phases"], leaves):
        g6["restated_grad_" + n] = t.grad.numpy()
    g6["restated_fwd_maxdiff"] = np.float64(fwd_diff)
    save(g6, "G6_phase256_128.npz")

    # G7 ---------------------------------------------------------------------------
    R = 96
    g = torch.Generator().manual_seed(7)
    N = 256
    pos7 = torch.randn(N, 3, generator=g) * 0.35
    scale7 = torch.rand(N, 3, generator=g) * 0.08 + 0.02
    rot7 = torch.randn(N, 4, generator=g)
    col7 = torch.rand(N, 3, generator=g)
    opa7 = torch.rand(N, generator=g)
    cam7 = frontal_camera(R)
    cam7.set_view(torch.from_numpy(orbit_view(20.0, 135.0)))
    g7 = run_tbr("G7", pos7, scale7, rot7, col7, opa7, cam7, R, R, bg=(0.05, 0.05, 0.05),
                 seed_up=107)
    save(g7, "G7_orbit256_96.npz")

    # G8 ---------------------------------------------------------------------------
    S = 64
    prop = AngularSpectrumPropagator(S, S, pixel_pitch=1.0 / 256.0, wavelength=0.05)
    rs = np.random.RandomState(8)
    f = (rs.standard_normal((S, S)) + 1j * rs.standard_normal((S, S))).astype(np.complex64)
    ft = torch.from_numpy(f)
    with torch.no_grad():
        out0 = prop.propagate(ft, torch.tensor(0.0))
        out1 = prop.propagate(ft, torch.tensor(0.3), torch.tensor(0.05))
        out2 = prop.propagate(ft, torch.tensor(-0.7), torch.tensor(0.0635))
        H1 = prop._compute_transfer_function(torch.tensor(0.3), torch.tensor(0.05))
    g8 = dict(field=f, out_z0=out0.numpy(), out_z03_l005=out1.numpy(),
              out_zm07_l00635=out2.numpy(), H_z03_l005=H1.numpy(),
              FX=prop.FX.numpy(), FY=prop.FY.numpy(), pixel_pitch=np.float64(1.0 / 256.0))
    for k, v in META.items():
        g8["meta_" + k] = np.array(v)
    save(g8, "G8_asm_propagator_64.npz")

    # G9 ---------------------------------------------------------------------------
    R = 128
    pos9, scale9, rot9, col9, opa9 = dummy_saag(256, 9)
    # ASM depth planes span 0.1..2.0; the SAAG cloud sits at depth ~2 +- 0.5 so several
    # planes are populated and the rest exercise the empty-plane skip.
    g = torch.Generator().manual_seed(9)
    ph_s = torch.rand(256, generator=g) * 2 * np.pi
    ph_v = torch.rand(256, 3, generator=g) * 2 * np.pi
    cam9 = frontal_camera(R)
    for tag, ph in (("scalar", ph_s), ("rgb", ph_v)):
        ren = ASMWaveFieldRenderer(R, R, background=(0.1, 0.1, 0.1))
        leaves = [t.clone().requires_grad_(True) for t in (pos9, scale9, rot9, col9, opa9, ph)]
        wl = torch.tensor([0.0635, 0.05, 0.041], requires_grad=True)
        img = ren(*leaves[:5], cam9, phases=leaves[5], wavelengths_rgb=wl)
        gI, _ = upstream(109, R, R)
        (img * torch.from_numpy(gI)).sum().backward()
        rec = dict(positions=pos9.numpy(), scales=scale9.numpy(), rotations=rot9.numpy(),
                   colors=col9.numpy(), opacities=opa9.numpy(), phases=ph.numpy(),
                   wavelengths=wl.detach().numpy(), background=np.array([0.1, 0.1, 0.1], np.float32),
                   view=cam9.view_matrix.numpy(), size=np.array([R, R], np.int32),
                   intr=np.array([cam9.fx, cam9.fy, cam9.cx, cam9.cy, cam9.near, cam9.far]),
                   gI=gI, seed_up=np.int32(109), image=img.detach().numpy(),
                   grad_wavelengths=wl.grad.numpy())
        for n, t in zip(["positions", "scales", "rotations", "colors", "opacities", "phases"], leaves):
            rec["grad_" + n] = t.grad.numpy()
        for k, v in META.items():
            rec["meta_" + k] = np.array(v)
        save(rec, f"G9_asm256_128_{tag}.npz")


def wave_goldens():
    """G10: WaveFieldRenderer (DR:689-926), N=256 @128x128, scalar and per-channel phases,
    image + depth map + all gradients."""
    R = 128
    pos, scale, rot, col, opa = dummy_saag(256, 10)
    g = torch.Generator().manual_seed(10)
    rot = torch.randn(256, 4, generator=g)
    scale = torch.rand(256, 3, generator=g) * 0.06 + 0.03
    ph_s = torch.rand(256, generator=g) * 2 * np.pi
    ph_v = torch.rand(256, 3, generator=g) * 2 * np.pi
    cam = frontal_camera(R)
    for tag, ph in (("scalar", ph_s), ("rgb", ph_v)):
        ren = WaveFieldRenderer(R, R, background=(0.1, 0.15, 0.2))
        leaves = [t.clone().requires_grad_(True) for t in (pos, scale, rot, col, opa, ph)]
        img, dep = ren(*leaves[:5], cam, return_depth=True, phases=leaves[5])
        gI, gD = upstream(110, R, R)
        ((img * torch.from_numpy(gI)).sum() + (dep * torch.from_numpy(gD)).sum()).backward()
        rec = dict(positions=pos.numpy(), scales=scale.numpy(), rotations=rot.numpy(), colors=col.numpy(),
                   opacities=opa.numpy(), phases=ph.numpy(), background=np.array([0.1, 0.15, 0.2], np.float32),
                   view=cam.view_matrix.numpy(), size=np.array([R, R], np.int32),
                   intr=np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.near, cam.far]), gI=gI, gD=gD,
                   seed_up=np.int32(110), image=img.detach().numpy(), depth=dep.detach().numpy())
        for n, t in zip(["positions", "scales", "rotations", "colors", "opacities", "phases"], leaves):
            rec["grad_" + n] = t.grad.numpy()
        for k, v in META.items():
            rec["meta_" + k] = np.array(v)
        save(rec, f"G10_wave256_128_{tag}.npz")


def loss_goldens():
    """G11.  scripts/training/train_gaussian_decoder.py imports torchvision at module level, which this image
    lacks, so the module cannot be imported whole.  Only the three definitions under test are taken from the
    reference file (parsed with `ast`, compiled and executed in a namespace that holds what they use: torch,
    nn, F, Tuple) -- the reference's own code runs, nothing of it is stored."""
    import ast
    import torch.nn as nn
    import torch.nn.functional as F
    from typing import Tuple
    path = os.path.join(REF, "training", "train_gaussian_decoder.py")
    tree = ast.parse(open(path).read())
    want = {"PhaseRetrievalLoss", "FrequencyDomainLoss", "wave_equation_loss"}
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in want]
    assert {n.name for n in body} == want
    ns = dict(torch=torch, nn=nn, F=F, Tuple=Tuple)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    rs = np.random.RandomState(1111)
    Bn, H, W = 2, 48, 40
    rendered = rs.uniform(0.0, 1.0, (Bn, 3, H, W)).astype(np.float32)
    rendered[0, :, :4, :4] = 0.0  # exercises the 1e-8 amplitude clamp
    target = rs.uniform(0.0, 1.0, (Bn, 3, H, W)).astype(np.float32)
    depth = rs.uniform(0.2, 3.0, (Bn, H, W)).astype(np.float32)
    rec = dict(rendered=rendered, target=target, depth=depth)

    def run(tag, fn, *arrs):
        ts = [torch.tensor(a, requires_grad=True) for a in arrs]
        loss = fn(*ts)
        loss.backward()
        rec[tag + "_loss"] = np.float32(loss.item())
        for i, t in enumerate(ts):
            rec[f"{tag}_grad{i}"] = t.grad.numpy()

    run("phase", lambda r, t, d: ns["PhaseRetrievalLoss"](wavelength=0.05, focal_depth=0.5)(r, t, d), rendered, target, depth)
    wl = torch.tensor(0.0635)
    run("phase_wl", lambda r, t, d: ns["PhaseRetrievalLoss"]()(r, t, d.unsqueeze(1), wavelength=wl), rendered, target, depth)
    run("freq", lambda r, t: ns["FrequencyDomainLoss"](cutoff=0.1, high_weight=2.0)(r, t), rendered, target)
    run("freq_c25", lambda r, t: ns["FrequencyDomainLoss"](cutoff=0.25, high_weight=0.5)(r, t), rendered, target)
    run("helm", lambda u: ns["wave_equation_loss"](u, 0.05), rendered)
    run("helm3", lambda u: ns["wave_equation_loss"](u, 0.0635, pixel_spacing=1.0 / 128.0), depth)
    for k, v in META.items():
        rec["meta_" + k] = np.array(v)
    save(rec, "G11_losses_48x40.npz")


def handoff_goldens():
    """G12: the caller-side pieces around the renderer, produced by the reference's own code.
    train_gaussian_decoder.py cannot be imported whole (torchvision), so -- as for G11 -- the definitions under test
    (HFTSConfig, create_camera_from_pose, ImageDataset) are parsed out of the file with `ast` and executed in a
    namespace holding what they use; rotate_positions_for_pose and load/save_gaussians_*_binary are imported from
    their modules.  Nothing of the reference's text is stored: only inputs (incl. the bytes of the small data files
    the dataset reads) and outputs."""
    import ast
    import io as _io
    import tempfile
    from dataclasses import dataclass
    from pathlib import Path
    from typing import Dict, Optional, Tuple
    from PIL import Image
    from torch.utils.data import Dataset
    import json
    import models.differentiable_renderer as DR
    from models.gaussian_decoder_models import rotate_positions_for_pose
    path = os.path.join(REF, "training", "train_gaussian_decoder.py")
    tree = ast.parse(open(path).read())
    want = {"HFTSConfig", "create_camera_from_pose", "ImageDataset"}
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in want]
    assert {n.name for n in body} == want
    ns = dict(torch=torch, np=np, dataclass=dataclass, Tuple=Tuple, Optional=Optional, Dict=Dict, Camera=DR.Camera,
              Dataset=Dataset, Path=Path, Image=Image, load_gaussians_from_binary=DR.load_gaussians_from_binary,
              transforms=None)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    rec = {}
    # ---- HFTSConfig tables (TGD:239-303)
    H = ns["HFTSConfig"]
    cfgs = [dict(), dict(progressive_schedule=True), dict(fast_mode=True), dict(stochastic_k=300), dict(train_resolution=96),
            dict(fast_mode=True, stochastic_k=100, train_resolution=128)]
    gpp, sk, tr = [], [], []
    for kw in cfgs:
        h = H(**kw)
        gpp.append([[h.get_gaussians_per_patch(e, T, b) for e in range(0, 21)] for (T, b) in [(20, 4), (20, 8), (7, 2), (0, 4)]])
        sk.append([-1 if h.get_stochastic_k(n) is None else h.get_stochastic_k(n) for n in (100, 256, 5476)])
        tr.append([h.get_effective_train_resolution(s) for s in (64, 256, 512)])
    rec["hfts_configs"] = np.array([json.dumps(kw) for kw in cfgs])
    rec["hfts_gpp"], rec["hfts_k"], rec["hfts_res"] = np.array(gpp), np.array(sk), np.array(tr)
    # ---- create_camera_from_pose (TGD:684-757)
    poses = [(0.0, 0.0), (20.0, 135.0), (-30.0, 270.0), (89.9999, 10.0), (90.0, 0.0), (45.0, 360.0)]
    views, intr = [], []
    for el, az in poses:
        cam = ns["create_camera_from_pose"](np.radians(el), np.radians(az), 96)
        views.append(cam.view_matrix.numpy())
        intr.append([cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height, cam.near, cam.far])
    cam = ns["create_camera_from_pose"](0.3, 1.1, 128, focal_length_mult=1.2, distance=3.5)
    views.append(cam.view_matrix.numpy()); intr.append([cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height, cam.near, cam.far])
    rec["pose_deg"], rec["pose_view"], rec["pose_intr"] = np.array(poses), np.array(views), np.array(intr)
    # ---- rotate_positions_for_pose (GDM:51-104)
    rs = np.random.RandomState(77)
    P = rs.standard_normal((3, 5, 5, 2, 3)).astype(np.float32)
    el, az = rs.uniform(-0.5, 0.8, 3).astype(np.float32), rs.uniform(0, 6.28, 3).astype(np.float32)
    rec["rot_in"], rec["rot_el"], rec["rot_az"] = P, el, az
    rec["rot_out"] = rotate_positions_for_pose(torch.from_numpy(P), torch.from_numpy(el), torch.from_numpy(az)).numpy()
    # ---- ImageDataset (TGD:525-675) over a tiny directory; the files' bytes travel with the fixture
    S, FD = 24, 4
    with tempfile.TemporaryDirectory() as td:
        files = {}
        def put(rel, data):
            full = os.path.join(td, rel)
            os.makedirs(os.path.dirname(full), exist_ok=True)
            open(full, "wb").write(data)
            files[rel] = np.frombuffer(data, dtype=np.uint8)
        for name, (w, h) in [("img_a", (28, 20)), ("img_b", (24, 24)), ("img_c", (9, 31))]:
            buf = _io.BytesIO()
            Image.fromarray(rs.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(buf, format="PNG")
            put(name + ".png", buf.getvalue())
        put("features/img_a_dinov2.bin", rs.standard_normal((37, 37, FD)).astype(np.float32).tobytes())
        put("features/img_a_depth.bin", rs.uniform(0, 1, (12, 12)).astype(np.float32).tobytes())   # resized 12 -> 24
        put("features/img_b_depth.bin", rs.uniform(0, 1, (S, S)).astype(np.float32).tobytes())     # native size
        saag = rs.standard_normal((5, 14)).astype(np.float32)
        put("features/img_b_saag.bin", saag.tobytes())
        ds = ns["ImageDataset"](td, image_size=S, use_augmentation=False, feature_dim=FD)
        assert len(ds) == 3
        for i in range(3):
            it = ds[i]
            for k in ("image", "features", "depth", "saag_positions", "saag_scales", "saag_rotations", "saag_colors", "saag_opacities"):
                rec[f"ds{i}_{k}"] = it[k].numpy()
            rec[f"ds{i}_name"], rec[f"ds{i}_has_saag"] = np.array(it["name"]), np.array(it["has_saag"])
        ds2 = ns["ImageDataset"](td, image_size=S, use_augmentation=False, max_images=2, feature_dim=768)
        rec["ds_max2_len"], rec["ds_suffix_768"] = np.array(len(ds2)), np.array(ds2.feature_suffix)
        rec["ds_files"] = np.array(sorted(files))
        for rel, data in files.items():
            rec["file:" + rel] = data
    rec["ds_image_size"], rec["ds_feature_dim"] = np.array(S), np.array(FD)
    for k, v in META.items():
        rec["meta_" + k] = np.array(v)
    save(rec, "G12_handoff.npz")


def midsize_golden():
    """G13: decoder-like grid made heavy enough that the 64-px radius cap is active and every pixel
    composites a few hundred Gaussians.  Image / depth rows 0::3 are stored (the gradients integrate all pixels)."""
    R, s = 256, 32
    N = s * s
    g = torch.Generator().manual_seed(13)
    lin = torch.linspace(-1.0, 1.0, s)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    pos = torch.stack([gx.reshape(-1), gy.reshape(-1), -2.0 - 2.0 * torch.rand(N, generator=g)], 1)
    scale = 0.2 + 0.25 * torch.rand(N, 3, generator=g)
    rot = torch.randn(N, 4, generator=g)
    col = torch.rand(N, 3, generator=g)
    opa = 0.02 + 0.1 * torch.rand(N, generator=g)   # low opacity: transmittance stays alive through the whole list
    rec = run_tbr("G13", pos, scale, rot, col, opa, frontal_camera(R), R, R, bg=(0.2, 0.1, 0.3), seed_up=113)
    vis = rec["visible"].astype(bool)
    bb = rec["bbox"][vis]
    pairs = int(((bb[:, 1] - bb[:, 0]) * (bb[:, 3] - bb[:, 2])).sum())
    print(f"G13: visible {int(vis.sum())}, capped {(rec['radii'][vis] >= 64).sum()}, pairs {pairs} = {pairs / R / R:.0f} per pixel")
    rec["rows"] = np.arange(0, R, 3, dtype=np.int32)
    rec["image"] = rec["image"][:, ::3].copy()
    rec["depth"] = rec["depth"][::3].copy()
    rec["pairs"] = np.int64(pairs)
    for k in ("gI", "gD"):   # regenerated from seed_up by the tests (upstream())
        rec.pop(k)
    save(rec, "G13_midsize1024_256.npz")


def config3_image_golden():
    """G15: the benchmark's own workload under the reference -- one image of config 3 (153 M composited Gaussian-pixels,
    ~2 minutes and ~12 GB of autograd graph here).  Stored: a subset (rows 0::16 of image / depth, the gradients of every
    8th Gaussian, the reference's integer stages packed) -- the inputs come from helpers.synth_saag(32768, 1503)."""
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from helpers import synth_saag
    R, N, seed = 512, 32768, 1503
    arrs = [torch.from_numpy(a) for a in synth_saag(N, seed)]
    # stable=True: ~130 pairs of the 32 768 fp32 depths tie, and the reference's default argsort is not stable
    rec = run_tbr("G15", *arrs, frontal_camera(R), R, R, bg=(0.0, 0.0, 0.0), seed_up=115, stable=True)
    vis = rec["visible"].astype(bool)
    bb = rec["bbox"][vis]
    pairs = int(((bb[:, 1] - bb[:, 0]) * (bb[:, 3] - bb[:, 2])).sum())
    print(f"G15: visible {int(vis.sum())}, capped {(rec['radii'][vis] >= 64).sum()}, pairs {pairs}")
    out = dict(name=np.array("G15"), seed=np.int32(seed), num_gaussians=np.int32(N), size=rec["size"], intr=rec["intr"],
               view=rec["view"], background=rec["background"], seed_up=rec["seed_up"], pairs=np.int64(pairs),
               rows=np.arange(0, R, 16, dtype=np.int32), image=rec["image"][:, ::16].copy(), depth=rec["depth"][::16].copy(),
               grad_stride=np.int32(8), visible=np.packbits(rec["visible"]), bbox=rec["bbox"].astype(np.int16),
               depth_order=rec["depth_order"].astype(np.int32))
    for n in ("positions", "scales", "rotations", "colors", "opacities"):
        out["grad_" + n] = rec["grad_" + n][::8].copy()
        out["gradmax_" + n] = np.float64(np.abs(rec["grad_" + n]).max())  # the tolerance is relative to the FULL tensor's max
    for k, v in META.items():
        out["meta_" + k] = np.array(v)
    save(out, "G15_config3_image_512.npz")


def _asm_reference_run(dt, arrs, phases, wl, gI, W, H, bg, **kw):
    """ASMWaveFieldRenderer itself with torch's default dtype set to `dt` (fp64: plane fields, FFTs, transfer functions
    and every gradient reduction in double; the reference's own `total_field` stays complex64, DR:1288)."""
    names = ["positions", "scales", "rotations", "colors", "opacities", "phases"]
    torch.set_default_dtype(dt)
    try:
        camd = Camera(fx=0.8 * W, fy=0.8 * W, cx=W / 2, cy=H / 2, width=W, height=H)
        camd.set_view(torch.eye(4, dtype=dt))
        ren = ASMWaveFieldRenderer(W, H, background=bg, **kw)
        leaves = [torch.from_numpy(a).to(dt).requires_grad_(True) for a in list(arrs) + [phases]]
        wlt = torch.from_numpy(wl).to(dt).requires_grad_(True)
        img = ren(*leaves[:5], camd, phases=leaves[5], wavelengths_rgb=wlt)
        assert img.dtype == dt, img.dtype
        (img * torch.from_numpy(gI).to(dt)).sum().backward()
    finally:
        torch.set_default_dtype(torch.float32)
    out = {"image": img.detach().numpy()}
    for n, t in zip(names, leaves):
        out["grad_" + n] = t.grad.numpy()
    out["grad_wavelengths"] = wlt.grad.numpy()
    return out


def config5_image_golden():
    """G16: ONE image of BASELINE config 5 as benchmarked (8 192 Gaussians @512x512, 16 depth planes 0.1..2.0, focal 0.5,
    pitch 1/256, per-channel wavelengths .0635/.05/.041, scalar phases U(0, 2 pi)) through the reference's
    ASMWaveFieldRenderer in fp32 AND in fp64, gradients incl. dL/dlambda.  Stored: image rows 0::16, the gradients of
    every 8th Gaussian (+ each FULL tensor's max, which the tolerance is relative to), dL/dlambda of both runs; inputs
    are regenerated by tests/helpers.synth_saag(8192, 1605) + RandomState(1606) phases.
    Also G9f64: the fp64 run of the two G9 scenes (the committed G9 fixtures hold the fp32 run only) -- the referee for
    dL/dlambda there."""
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from helpers import synth_saag
    names = ["positions", "scales", "rotations", "colors", "opacities", "phases"]
    # ---- G9 in fp64 ----
    for tag in ("scalar", "rgb"):
        g9 = np.load(os.path.join(OUT, f"G9_asm256_128_{tag}.npz"))
        R = int(g9["size"][0])
        arrs = [g9[n] for n in names[:5]]
        r64 = _asm_reference_run(torch.float64, arrs, g9["phases"], g9["wavelengths"], g9["gI"], R, R,
                                 tuple(float(v) for v in g9["background"]))
        rec = {"f64_" + k: v for k, v in r64.items()}
        r32 = _asm_reference_run(torch.float32, arrs, g9["phases"], g9["wavelengths"], g9["gI"], R, R,
                                 tuple(float(v) for v in g9["background"]))
        # (a re-run of the committed fp32 fixture: equal up to the summation order of torch's threaded reductions)
        d_img = float(np.abs(r32["image"] - g9["image"]).max())
        fin = np.isfinite(g9["grad_wavelengths"])  # (lambda = 0.05 puts frequencies ON the evanescent boundary: the
        assert np.array_equal(fin, np.isfinite(r32["grad_wavelengths"]))  # reference's fp32 autograd returns NaN there)
        d_wl = float(np.abs(r32["grad_wavelengths"] - g9["grad_wavelengths"])[fin].max() / np.abs(g9["grad_wavelengths"][fin]).max())
        print(f"G9 {tag}: fp32 re-run vs committed fixture: image {d_img:.1e} abs, dL/dlambda {d_wl:.1e} rel")
        assert d_img <= 1e-6 and d_wl <= 1e-4, (d_img, d_wl)
        rec["f32rerun_grad_wavelengths"] = r32["grad_wavelengths"]
        rec["f64_image"] = rec["f64_image"].astype(np.float32)
        rec["case"] = np.array(f"G9_asm256_128_{tag}")
        for k, v in META.items():
            rec["meta_" + k] = np.array(v)
        print(f"G9f64 {tag}: reference fp32 vs fp64 (rel to max): " + ", ".join(
            f"{k} {np.nanmax(np.abs(r32[k] - r64[k])) / np.abs(r64[k]).max():.1e}" for k in r64)
              + f"; dL/dlambda fp32 {r32['grad_wavelengths']} fp64 {r64['grad_wavelengths']}")
        save(rec, f"G9f64_asm256_128_{tag}.npz")
    # ---- G16 ----
    S, N, seed = 512, 8192, 1605
    arrs = list(synth_saag(N, seed))
    phases = (np.random.RandomState(seed + 1).random_sample(N) * 2 * np.pi).astype(np.float32)
    wl = np.array([0.0635, 0.05, 0.041], np.float32)
    gI, _ = upstream(116, S, S)
    out = dict(name=np.array("G16"), seed=np.int32(seed), num_gaussians=np.int32(N), size=np.array([S, S], np.int32),
               wavelengths=wl, background=np.zeros(3, np.float32), seed_up=np.int32(116), grad_stride=np.int32(8),
               rows=np.arange(0, S, 16, dtype=np.int32), num_depth_planes=np.int32(16), depth_range=np.array([0.1, 2.0]),
               focal_depth=np.float64(0.5), pixel_pitch=np.float64(1.0 / 256.0))
    runs = {}
    for dt, pre in ((torch.float32, "f32_"), (torch.float64, "f64_")):
        import time
        t0 = time.time()
        r = runs[pre] = _asm_reference_run(dt, arrs, phases, wl, gI, S, S, (0.0, 0.0, 0.0))
        print(f"G16 {pre}: {time.time() - t0:.0f} s, dL/dlambda = {r['grad_wavelengths']}")
        out[pre + "image"] = r["image"][:, ::16].astype(np.float32).copy()
        out[pre + "grad_wavelengths"] = r["grad_wavelengths"]
        for n in names:
            out[pre + "grad_" + n] = r["grad_" + n][::8].copy()
            out[pre + "gradmax_" + n] = np.float64(np.abs(r["grad_" + n]).max())
    print("G16 reference fp32 vs fp64 (rel to max): " + ", ".join(
        f"{k} {np.nanmax(np.abs(runs['f32_'][k] - runs['f64_'][k])) / np.abs(runs['f64_'][k]).max():.1e}" for k in runs["f64_"]))
    for k, v in META.items():
        out["meta_" + k] = np.array(v)
    save(out, "G16_config5_image_512.npz")


def needle_goldens():
    """G14: needles (s, s/r, s/r) and discs (s, s, s/r) at ratio r, reference in fp32 and in fp64.  The fp64 run is
    the reference's own code with torch's default dtype set to float64 and double inputs / view matrix."""
    R, N = 96, 48
    for ratio in (30, 100, 500):
        g = torch.Generator().manual_seed(1400 + ratio)
        pos = torch.randn(N, 3, generator=g) * 0.4
        pos[:, 2] -= 2
        smax = 0.1 + 0.3 * torch.rand(N, generator=g)
        scale = torch.stack([smax, smax / ratio, smax / ratio], 1)
        scale[N // 2:, 1] = smax[N // 2:]            # second half: discs
        perm = torch.stack([torch.randperm(3, generator=g) for _ in range(N)])
        scale = torch.gather(scale, 1, perm)         # the thin axis is not always the same one
        rot = torch.randn(N, 4, generator=g)
        col = torch.rand(N, 3, generator=g)
        opa = 0.2 + 0.8 * torch.rand(N, generator=g)
        cam = frontal_camera(R)
        rec = run_tbr(f"G14_r{ratio}", pos, scale, rot, col, opa, cam, R, R, bg=(0.1, 0.1, 0.1), seed_up=114)
        torch.set_default_dtype(torch.float64)
        try:
            cam64 = frontal_camera(R)
            cam64.set_view(torch.eye(4, dtype=torch.float64))
            ren = TileBasedRenderer(R, R, background=(0.1, 0.1, 0.1))
            leaves = [t.double().clone().requires_grad_(True) for t in (pos, scale, rot, col, opa)]
            img, dep = ren(*leaves, cam64, return_depth=True)
            assert img.dtype == torch.float64
            ((img * torch.from_numpy(rec["gI"]).double()).sum() + (dep * torch.from_numpy(rec["gD"]).double()).sum()).backward()
            inter64 = intermediates(ren, *[t.detach() for t in leaves[:3]], cam64, False)
        finally:
            torch.set_default_dtype(torch.float32)
        same_sets = bool(np.array_equal(inter64["visible"], rec["visible"]) and np.array_equal(inter64["bbox"], rec["bbox"])
                         and np.array_equal(inter64["depth_order"], rec["depth_order"]))
        rec["f64_same_integer_stages"] = np.uint8(same_sets)
        rec["f64_image"], rec["f64_depth"] = img.detach().numpy().astype(np.float32), dep.detach().numpy().astype(np.float32)
        worst = {}
        for n, t in zip(["positions", "scales", "rotations", "colors", "opacities"], leaves):
            rec["f64_grad_" + n] = t.grad.numpy()
            m = np.abs(rec["f64_grad_" + n]).max()
            worst[n] = float(np.abs(rec["grad_" + n] - rec["f64_grad_" + n]).max() / m)
        rec["scale_ratio"] = np.int32(ratio)
        for k in ("gI", "gD"):   # regenerated from seed_up by the tests
            rec.pop(k)
        print(f"G14 ratio {ratio}: same integer stages in fp64: {same_sets}; reference fp32 vs fp64 (rel to max): "
              + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
        save(rec, f"G14_needles_r{ratio}_96.npz")


def kink_goldens():
    """K1-K4.  Round-2 sweeps (HIP vs this repo's oracle) found four cases above 1e-4; each is replayed here from its
    (seed, iteration) and given a referee from the reference's own code in fp32 AND fp64:
      phase path (K1 = fuzz_phase seed 2 it 12, K2 = seed 1 it 23): the reference forward (it cannot backprop this
        path) + the out-of-place restatement of G6, which must reproduce that forward exactly;
      ASM (K3 = fuzz_asm seed 3 it 10, K4 = seed 5 it 8): ASMWaveFieldRenderer itself."""
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from fuzz_cases import asm_cases, phase_cases
    names = ["positions", "scales", "rotations", "colors", "opacities", "phases"]
    for tag, seed, it in (() if "--k6-only" in sys.argv else (("K1", 2, 12), ("K2", 1, 23))):
        c = [c for c in phase_cases(seed) if c["it"] == it][0]
        W, H, amp, bg = c["W"], c["H"], c["amp"], c["bg"]
        ts = [torch.from_numpy(a) for a in c["arrs"]]
        ph = torch.from_numpy(c["phases"])
        cam = Camera(fx=0.8 * W, fy=0.8 * W, cx=W / 2, cy=H / 2, width=W, height=H)
        ref = run_tbr(tag, *ts, cam, H, W, bg=bg, phases=ph, use_phase=True, amp=amp, stable=True, grad_inputs="color")
        rec = {k: ref[k] for k in ("positions", "scales", "rotations", "colors", "opacities", "phases", "view", "intr", "size",
                                   "background", "image", "depth", "phase_amplitude", "visible", "bbox", "depth_order")}
        rec["gI"], rec["gD"] = c["gI"], c["gD"]
        for dt, pre in ((torch.float32, "f32_"), (torch.float64, "f64_")):
            torch.set_default_dtype(dt)
            try:
                camd = Camera(fx=0.8 * W, fy=0.8 * W, cx=W / 2, cy=H / 2, width=W, height=H)
                camd.set_view(torch.eye(4, dtype=dt))
                leaves = [t.to(dt).clone().requires_grad_(True) for t in ts + [ph]]
                img, dep = phase_restatement(*leaves, camd, H, W, bg, amp, ref)
                if dt == torch.float32:
                    d = max(float((img.detach() - torch.from_numpy(ref["image"])).abs().max()),
                            float((dep.detach() - torch.from_numpy(ref["depth"])).abs().max()))
                    print(f"{tag}: restatement vs reference forward: max abs diff = {d}")
                    assert d == 0.0
                ((img * torch.from_numpy(c["gI"]).to(dt)).sum() + (dep * torch.from_numpy(c["gD"]).to(dt)).sum()).backward()
            finally:
                torch.set_default_dtype(torch.float32)
            if dt == torch.float64:   # (the fp32 restatement's forward IS rec["image"] / rec["depth"], asserted above)
                rec[pre + "image"], rec[pre + "depth"] = img.detach().numpy().astype(np.float32), dep.detach().numpy().astype(np.float32)
            for n, t in zip(names, leaves):
                rec[pre + "grad_" + n] = t.grad.numpy()
        for k, v in META.items():
            rec["meta_" + k] = np.array(v)
        rec["sweep"] = np.array(f"fuzz_phase seed {seed} it {it}")
        print(tag, {n: f"{np.abs(rec['f32_grad_' + n] - rec['f64_grad_' + n]).max() / np.abs(rec['f64_grad_' + n]).max():.1e}" for n in names})
        save(rec, f"{tag}_phase_kink_s{seed}_it{it}.npz")
    asm_kinks = (("K3", 3, 10), ("K4", 5, 8), ("K5", 8, 0),  # K5: found by the round-3 sweep on the same build
                 ("K6", 0, 4))  # K6 (round 5): the one case of the 840-case sweep outside both rules -- a cancelling moment sum, not a kink
    if "--k6-only" in sys.argv:
        asm_kinks = asm_kinks[3:]
    for tag, seed, it in asm_kinks:
        c = [c for c in asm_cases(seed) if c["it"] == it][0]
        assert c["kind"] == "asm"
        W, H, bg, kw = c["W"], c["H"], c["bg"], c["kw"]
        rec = dict(zip(names[:5], c["arrs"]))
        rec.update(phases=c["phases"], wavelengths=c["wl"], background=np.array(bg, np.float32), size=np.array([W, H], np.int32),
                   gI=c["gI"], num_depth_planes=np.int32(kw["num_depth_planes"]), depth_range=np.array(kw["depth_range"]),
                   focal_depth=np.float64(kw["focal_depth"]), pixel_pitch=np.float64(kw["pixel_pitch"]),
                   intr=np.array([0.8 * W, 0.8 * W, W / 2, H / 2, 0.01, 100.0]), view=np.eye(4, dtype=np.float32))
        for dt, pre in ((torch.float32, "f32_"), (torch.float64, "f64_")):
            torch.set_default_dtype(dt)
            try:
                camd = Camera(fx=0.8 * W, fy=0.8 * W, cx=W / 2, cy=H / 2, width=W, height=H)
                camd.set_view(torch.eye(4, dtype=dt))
                ren = ASMWaveFieldRenderer(W, H, background=bg, **kw)
                leaves = [torch.from_numpy(a).to(dt).requires_grad_(True) for a in c["arrs"] + [c["phases"]]]
                wl = torch.from_numpy(c["wl"]).to(dt).requires_grad_(True)
                img = ren(*leaves[:5], camd, phases=leaves[5], wavelengths_rgb=wl)
                assert img.dtype == dt, img.dtype
                (img * torch.from_numpy(c["gI"]).to(dt)).sum().backward()
            finally:
                torch.set_default_dtype(torch.float32)
            rec[pre + "image"] = img.detach().numpy().astype(np.float32)
            for n, t in zip(names, leaves):
                rec[pre + "grad_" + n] = t.grad.numpy()
            rec[pre + "grad_wavelengths"] = wl.grad.numpy()
        rec["sweep"] = np.array(f"fuzz_asm seed {seed} it {it}")
        print(tag, {n: f"{np.abs(rec['f32_grad_' + n] - rec['f64_grad_' + n]).max() / max(np.abs(rec['f64_grad_' + n]).max(), 1e-300):.1e}" for n in names})
        for k, v in META.items():
            rec["meta_" + k] = np.array(v)
        save(rec, f"{tag}_asm_kink_s{seed}_it{it}.npz")


if __name__ == "__main__":
    if "--config3-only" in sys.argv:
        config3_image_golden()
        config5_image_golden()
    elif "--config5-only" in sys.argv:
        config5_image_golden()
    elif "--kinks-only" in sys.argv or "--k6-only" in sys.argv:
        kink_goldens()
    elif "--midsize-only" in sys.argv:
        midsize_golden()
    elif "--needles-only" in sys.argv:
        needle_goldens()
    elif "--handoff-only" in sys.argv:
        handoff_goldens()
    elif "--wave-only" in sys.argv:
        wave_goldens()
    elif "--losses-only" in sys.argv:
        loss_goldens()
    else:
        main()
        wave_goldens()
        loss_goldens()
        handoff_goldens()
        midsize_golden()
        needle_goldens()
        kink_goldens()
        config3_image_golden()
        config5_image_golden()
