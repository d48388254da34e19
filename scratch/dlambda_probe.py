"""dL/dlambda of the HIP ASM renderer against the reference's fp32 and fp64 runs (G9 + G9f64, K3-K5, G16): which side loses
the digits.  GPU box:  python scratch/dlambda_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, rel_to_max, synth_saag, upstream_grads  # noqa: E402
from fresnel_amd.renderer import ASMWaveFieldRenderer, Camera  # noqa: E402

dev = torch.device("cuda:0")
NAMES = ["positions", "scales", "rotations", "colors", "opacities"]


def hip(arrs, phases, wl, W, H, bg, gI, batch_pad=0, **kw):
    """Renders the scene as image 0 of a batch of 1 + batch_pad images (the pad images are other saag scenes)."""
    ts = [torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(True) for a in arrs]
    ph = torch.from_numpy(phases).to(dev).requires_grad_(True)
    wlt = torch.from_numpy(np.asarray(wl, np.float32)).to(dev).requires_grad_(True)
    cam = Camera(0.8 * W, 0.8 * W, W / 2, H / 2, W, H)
    ren = ASMWaveFieldRenderer(W, H, background=tuple(float(b) for b in bg), **kw).to(dev)
    img = ren(*ts, cam, phases=ph, wavelengths_rgb=wlt)
    (img * torch.from_numpy(gI).to(dev)).sum().backward()
    out = {"grad_" + n: t.grad.cpu().numpy() for n, t in zip(NAMES, ts)}
    out.update(image=img.detach().cpu().numpy(), grad_phases=ph.grad.cpu().numpy(), grad_wavelengths=wlt.grad.cpu().numpy())
    return out


def report(tag, out_wl, ref32, ref64):
    m = np.abs(ref64).max()
    fin = np.isfinite(ref32)
    print(f"{tag}: dL/dlambda HIP {out_wl}  ref32 {ref32}  ref64 {ref64}")
    print(f"    HIP vs ref64 (rel to max): {np.abs(out_wl - ref64) / m}   ref32 vs ref64: {np.abs(ref32 - ref64)[fin] / m} (finite channels)")


for tag in ("scalar", "rgb"):
    g, f = load_golden(f"G9_asm256_128_{tag}"), load_golden(f"G9f64_asm256_128_{tag}")
    W, H = [int(v) for v in g["size"]]
    o = hip([g[k] for k in NAMES], g["phases"], g["wavelengths"], W, H, g["background"], g["gI"])
    report("G9 " + tag, o["grad_wavelengths"], g["grad_wavelengths"], f["f64_grad_wavelengths"])
    for k in NAMES + ["phases"]:
        print(f"    grad_{k}: HIP vs ref32 {rel_to_max(o['grad_' + k], g['grad_' + k]):.1e}  vs ref64 {rel_to_max(o['grad_' + k], f['f64_grad_' + k]):.1e}"
              f"  ref32 vs ref64 {rel_to_max(g['grad_' + k], f['f64_grad_' + k]):.1e}")
for name in ("K3_asm_kink_s3_it10", "K4_asm_kink_s5_it8", "K5_asm_kink_s8_it0"):
    g = load_golden(name)
    W, H = [int(v) for v in g["size"]]
    kw = dict(num_depth_planes=int(g["num_depth_planes"]), depth_range=tuple(float(v) for v in g["depth_range"]),
              focal_depth=float(g["focal_depth"]), pixel_pitch=float(g["pixel_pitch"]))
    o = hip([g[k] for k in NAMES], g["phases"], g["wavelengths"], W, H, g["background"], g["gI"], **kw)
    report(name[:2], o["grad_wavelengths"], g["f32_grad_wavelengths"], g["f64_grad_wavelengths"])
p16 = os.path.join(ROOT, "tests", "golden", "G16_config5_image_512.npz")
if os.path.exists(p16):
    g = np.load(p16)
    S, N, seed = 512, int(g["num_gaussians"]), int(g["seed"])
    arrs = list(synth_saag(N, seed))
    phases = (np.random.RandomState(seed + 1).random_sample(N) * 2 * np.pi).astype(np.float32)
    gI, _ = upstream_grads(int(g["seed_up"]), S, S)
    o = hip(arrs, phases, g["wavelengths"], S, S, (0.0, 0.0, 0.0), gI)
    report("G16 (b1)", o["grad_wavelengths"], g["f32_grad_wavelengths"], g["f64_grad_wavelengths"])
    st = int(g["grad_stride"])
    print("    image rows: HIP vs ref32", np.abs(o["image"][:, ::16] - g["f32_image"]).max(), " vs ref64", np.abs(o["image"][:, ::16] - g["f64_image"]).max())
    for k in NAMES + ["phases"]:
        m32, m64 = float(g["f32_gradmax_" + k]), float(g["f64_gradmax_" + k])
        print(f"    grad_{k}: HIP vs ref32 {np.abs(o['grad_' + k][::st] - g['f32_grad_' + k]).max() / m32:.1e}  vs ref64 "
              f"{np.abs(o['grad_' + k][::st] - g['f64_grad_' + k]).max() / m64:.1e}  ref32 vs ref64 "
              f"{np.abs(g['f32_grad_' + k] - g['f64_grad_' + k]).max() / m64:.1e}")
