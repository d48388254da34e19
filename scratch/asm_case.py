"""one column-fused ASM case against the oracle: asm_case.py W H P [Bn]; BYPLANE=1 / BYX=1 / KINK=1 / REFEREE=1 print where a
discrepancy sits (by depth plane, by screen position, saturated pixels and arg-max, oracle fp32 vs fp64).  Scratch diagnostic: the
352 x 64 x 6 planes x 2 images case of round 3 (a pixel whose summed amplitude sits within 1e-6 of the clamp at 1, DR:1327) was found with it."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import rel_to_max, synth_aniso
import test_hip_asm as T
from oracle import asm_oracle, fgs_oracle as orc
from fresnel_amd.renderer import Camera
W, H, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Bn = int(sys.argv[4]) if len(sys.argv) > 4 else 2
N = 300
bg = (0.05, 0.1, 0.15)
rs = np.random.RandomState(W + H)
per = []
for b in range(Bn):
    pos, scale, quat, col, opa = synth_aniso(N, 170 + b, opacity_max=0.9, smin=0.03, smax=0.1)
    pos[:, 1] *= H / W * 0.6 if H > W else 1.0
    pos[:, 2] = -rs.uniform(0.3, 2.0, N).astype(np.float32)
    per.append((pos, scale, quat, col, opa))
if os.environ.get("SWAP"): per = per[::-1]
arrs = [np.stack([p[i] for p in per]) for i in range(5)]
phases = (rs.random_sample((Bn, N, 3)) * 2 * np.pi).astype(np.float32)
wl = np.array([0.07, 0.052, 0.043], np.float32)
gI = rs.standard_normal((Bn, 3, H, W)).astype(np.float32)
f = 0.8 * min(W, H)
cam = Camera(f, f, W / 2, H / 2, W, H)
kw = dict(num_depth_planes=P, depth_range=(0.3, 2.2), focal_depth=0.9, pixel_pitch=1.0 / 200.0)
out = T._hip_asm(arrs, phases, wl, cam, W, H, bg, gI=gI, **kw)
ocam = orc.make_camera(np.eye(4, dtype=np.float32), f, f, W / 2, H / 2, W, H)
gw = 0.0
for b in range(Bn):
    r = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=bg, num_planes=P, depth_range=(0.3, 2.2), focal_depth=0.9,
                          pixel_pitch=1.0 / 200.0, grad_out=gI[b])
    line = ["img %.1e" % np.abs(out["image"][b] - r["image"]).max()]
    for k in ["positions", "scales", "rotations", "colors", "opacities", "phases"]:
        line.append("%s %.1e" % (k[:3], rel_to_max(out["grad_" + k][b], r["grad_" + k])))
    print(W, H, P, "image", b, " ".join(line), "planes used", sorted(set(r["plane_idx"].tolist())))
    gw = gw + r["grad_wavelengths"]
print("wavelengths", rel_to_max(out["grad_wavelengths"], gw))
if os.environ.get("BYPLANE"):
    b = 0
    r = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=bg, num_planes=P, depth_range=(0.3, 2.2), focal_depth=0.9,
                          pixel_pitch=1.0 / 200.0, grad_out=gI[b])
    pi = r["plane_idx"]
    g, o = out["grad_colors"][b], r["grad_colors"]
    for p in sorted(set(pi.tolist())):
        m = pi == p
        print("plane", p, "n", int(m.sum()), "max err", np.abs(g[m] - o[m]).max(), "max ref", np.abs(o[m]).max(), "per channel err", np.abs(g[m] - o[m]).max(axis=0))
if os.environ.get("KINK"):
    b = 0
    r = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=bg, num_planes=P, depth_range=(0.3, 2.2), focal_depth=0.9,
                          pixel_pitch=1.0 / 200.0, grad_out=gI[b])
    for name, im in (("hip", out["image"][b]), ("oracle", r["image"])):
        print(name, "pixels == 1.0:", int((im == 1.0).sum()), " in (1 - 3e-7, 1):", int(((im > 1 - 3e-7) & (im < 1.0)).sum()), " == 0:", int((im == 0.0).sum()),
              " max", repr(float(im.max())))
    d = (out["image"][b] == 1.0) != (r["image"] == 1.0)
    print("pixels saturated in one and not the other:", int(d.sum()), "|gI| there:", np.abs(gI[b])[d])
if os.environ.get("KINK"):
    for name, im in (("hip", out["image"][0]), ("oracle", r["image"])):
        flat = im.reshape(-1)
        top = np.argsort(flat)[-3:][::-1]
        print(name, "top-3 (index, value):", [(int(i), repr(float(flat[i]))) for i in top])
if os.environ.get("REFEREE"):
    import torch
    b = 0
    r32 = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=bg, num_planes=P, depth_range=(0.3, 2.2), focal_depth=0.9,
                            pixel_pitch=1.0 / 200.0, grad_out=gI[b])
    r64 = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=bg, num_planes=P, depth_range=(0.3, 2.2), focal_depth=0.9,
                            pixel_pitch=1.0 / 200.0, grad_out=gI[b], dtype=torch.float64)
    for k in ["colors", "opacities", "phases"]:
        print(k, "oracle fp32 vs fp64: %.2e   hip vs fp64: %.2e   hip vs fp32: %.2e" % (
            rel_to_max(r32["grad_" + k], r64["grad_" + k]), rel_to_max(out["grad_" + k][b], r64["grad_" + k]), rel_to_max(out["grad_" + k][b], r32["grad_" + k])))
if os.environ.get("BYX"):
    b = 0
    r = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=bg, num_planes=P, depth_range=(0.3, 2.2), focal_depth=0.9,
                          pixel_pitch=1.0 / 200.0, grad_out=gI[b])
    mx = r["proj"]["mean2d"][:, 0]; my = r["proj"]["mean2d"][:, 1]
    g, o = out["grad_colors"][b], r["grad_colors"]
    err = np.abs(g - o).max(axis=1)
    order = np.argsort(err)[::-1][:12]
    for i in order: print("gaussian %3d x %.1f y %.1f plane %d radius %.1f err %.4f ref %.3f" % (i, mx[i], my[i], r["plane_idx"][i], r["proj"]["radius"][i] if "radius" in r["proj"] else -1, err[i], np.abs(o[i]).max()))
    bins = np.linspace(0, W, 12)
    for lo, hi in zip(bins[:-1], bins[1:]):
        m = (mx >= lo) & (mx < hi)
        if m.any(): print("x in [%3.0f, %3.0f): n %3d max err %.4f mean err %.5f" % (lo, hi, m.sum(), err[m].max(), err[m].mean()))
