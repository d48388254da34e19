"""Referee table: for every fixture that holds the reference in fp32 AND fp64 (G14 needles, K1-K5 kinks, G9 + G9f64,
G16) the distance, relative to the tensor's max, of (a) the reference's own fp32 run, (b) this repo's CPU oracle, (c) the HIP path
from the fp64 reference.  GPU box:  python scratch/referee_table.py > profiles/r04_referee_table.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, oracle_camera, rel_to_max, upstream_grads  # noqa: E402
from oracle import asm_oracle, fgs_oracle as orc  # noqa: E402  (the checker, as in tests/)
import test_hip_parity as thp  # noqa: E402
import test_hip_asm as tha  # noqa: E402

NAMES = ["positions", "scales", "rotations", "colors", "opacities"]
rows = []


def add(case, tensor, ref32, ref64, oracle, hip):
    fin = np.isfinite(np.asarray(ref32, dtype=np.float64))
    sp = rel_to_max(np.asarray(ref32)[fin], np.asarray(ref64)[fin]) if fin.any() else float("nan")
    rows.append((case, tensor, sp, rel_to_max(oracle, ref64) if oracle is not None else float("nan"), rel_to_max(hip, ref64)))


def arrs(g):
    return [g[k] for k in NAMES]


for ratio in (30, 100, 500):
    g = load_golden(f"G14_needles_r{ratio}_96")
    W, H = [int(v) for v in g["size"]]
    gI, gD = upstream_grads(int(g["seed_up"]), H, W)
    r = orc.render(*arrs(g), oracle_camera(g), bg=tuple(float(b) for b in g["background"]))
    go = orc.render_backward(r, gI, gD)
    out = thp._hip_render(arrs(g), thp._camera_from_golden(g), W, H, g["background"], grads=(gI, gD))
    add(f"G14 r{ratio}", "image", g["image"], g["f64_image"], r.image, out["image"])
    for k in NAMES:
        add(f"G14 r{ratio}", k, g["grad_" + k], g["f64_grad_" + k], go[k], out["grad_" + k])
for case in ("K1_phase_kink_s2_it12", "K2_phase_kink_s1_it23"):
    g = load_golden(case)
    W, H = [int(v) for v in g["size"]]
    amp = float(g["phase_amplitude"])
    r = orc.render(*arrs(g), oracle_camera(g), bg=tuple(float(b) for b in g["background"]), phases=g["phases"], phase_amp=amp)
    go = orc.render_backward(r, g["gI"], g["gD"])
    out = thp._hip_render(arrs(g), thp._camera_from_golden(g), W, H, g["background"], phases=g["phases"], use_phase=True, amp=amp,
                          grads=(g["gI"], g["gD"]))
    add(case[:2], "image", g["image"], g["f64_image"], r.image, out["image"])
    for k in NAMES + ["phases"]:
        add(case[:2], k, g["f32_grad_" + k], g["f64_grad_" + k], go[k], out["grad_" + k])
for case in ("K3_asm_kink_s3_it10", "K4_asm_kink_s5_it8", "K5_asm_kink_s8_it0"):
    g = load_golden(case)
    W, H = [int(v) for v in g["size"]]
    kw = dict(num_depth_planes=int(g["num_depth_planes"]), depth_range=tuple(float(v) for v in g["depth_range"]),
              focal_depth=float(g["focal_depth"]), pixel_pitch=float(g["pixel_pitch"]))
    r = asm_oracle.render(*arrs(g), g["phases"], g["wavelengths"], oracle_camera(g), grad_out=g["gI"], bg=tuple(float(b) for b in g["background"]),
                          num_planes=kw["num_depth_planes"], depth_range=kw["depth_range"], focal_depth=kw["focal_depth"], pixel_pitch=kw["pixel_pitch"])
    out = tha._hip_asm(arrs(g), g["phases"], g["wavelengths"], tha._cam(g), W, H, g["background"], gI=g["gI"], **kw)
    add(case[:2], "image", g["f32_image"], g["f64_image"], r["image"], out["image"])
    for k in NAMES + ["phases", "wavelengths"]:
        add(case[:2], k, g["f32_grad_" + k], g["f64_grad_" + k], np.asarray(r["grad_" + k]), out["grad_" + k])
for tag in ("scalar", "rgb"):
    g, f = load_golden(f"G9_asm256_128_{tag}"), load_golden(f"G9f64_asm256_128_{tag}")
    W, H = [int(v) for v in g["size"]]
    r = asm_oracle.render(*arrs(g), g["phases"], g["wavelengths"], oracle_camera(g), grad_out=g["gI"], bg=tuple(float(b) for b in g["background"]))
    out = tha._hip_asm(arrs(g), g["phases"], g["wavelengths"], tha._cam(g), W, H, g["background"], gI=g["gI"])
    add("G9 " + tag, "image", g["image"], f["f64_image"], r["image"], out["image"])
    for k in NAMES + ["phases", "wavelengths"]:
        o = np.asarray(r["grad_" + k], dtype=np.float64)
        if k == "wavelengths":
            o = np.where(np.isfinite(o), o, f["f64_grad_" + k])  # (the fp32 oracle is NaN where the reference's fp32 run is)
        add("G9 " + tag, k, g["grad_" + k], f["f64_grad_" + k], o, out["grad_" + k])
print("distance from the reference's fp64 run, relative to the tensor's max")
print(f"{'case':10s} {'tensor':12s} {'reference fp32':>15s} {'CPU oracle':>12s} {'HIP':>12s} {'HIP / ref fp32':>15s}")
for case, t, sp, o, h in rows:
    print(f"{case:10s} {t:12s} {sp:15.2e} {o:12.2e} {h:12.2e} {h / sp if sp > 0 else float('nan'):15.2f}")
