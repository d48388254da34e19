"""how far the HIP ASM forward is from the fp32 and the fp64 oracle (image, max abs) on a few batched column-kernel shapes"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import rel_to_max, synth_aniso
from oracle import asm_oracle, fgs_oracle as orc
from fresnel_amd.renderer import Camera
import test_hip_asm as T
for (W, H, P, Bn, N, seed) in [(352, 64, 6, 2, 300, 1), (304, 128, 16, 2, 300, 2), (256, 256, 16, 1, 150, 3), (96, 512, 16, 1, 240, 4)]:
    rs = np.random.RandomState(seed)
    per = []
    for b in range(Bn):
        a = list(synth_aniso(N, 900 + 10 * seed + b, opacity_max=0.9, smin=0.03, smax=0.1))
        if H > W: a[0][:, 1] *= H / W * 0.6
        a[0][:, 2] = -rs.uniform(0.3, 2.9, N).astype(np.float32)
        per.append(a)
    arrs = [np.stack([p[i] for p in per]) for i in range(5)]
    phases = (rs.random_sample((Bn, N, 3)) * 2 * np.pi).astype(np.float32)
    wl = np.array([0.07, 0.052, 0.043], np.float32)
    gI = rs.standard_normal((Bn, 3, H, W)).astype(np.float32)
    f = 0.8 * min(W, H)
    cam = Camera(f, f, W / 2, H / 2, W, H)
    kw = dict(num_depth_planes=P, depth_range=(0.3, 3.0), focal_depth=0.9, pixel_pitch=1.0 / 200.0)
    out = T._hip_asm(arrs, phases, wl, cam, W, H, (0.05, 0.1, 0.15), gI=gI, **kw)
    ocam = orc.make_camera(np.eye(4, dtype=np.float32), f, f, W / 2, H / 2, W, H)
    e32 = e64 = r3264 = g32 = g64 = gr = 0.0
    for b in range(Bn):
        r32 = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=(0.05, 0.1, 0.15), num_planes=P, depth_range=(0.3, 3.0), focal_depth=0.9, pixel_pitch=1.0 / 200.0, grad_out=gI[b])
        r64 = asm_oracle.render(*[a[b] for a in arrs], phases[b], wl, ocam, bg=(0.05, 0.1, 0.15), num_planes=P, depth_range=(0.3, 3.0), focal_depth=0.9, pixel_pitch=1.0 / 200.0, grad_out=gI[b], dtype=torch.float64)
        e32 = max(e32, np.abs(out["image"][b] - r32["image"]).max()); e64 = max(e64, np.abs(out["image"][b] - r64["image"]).max())
        r3264 = max(r3264, np.abs(r32["image"] - r64["image"]).max())
        g32 = max(g32, rel_to_max(out["grad_colors"][b], r32["grad_colors"])); g64 = max(g64, rel_to_max(out["grad_colors"][b], r64["grad_colors"]))
        gr = max(gr, rel_to_max(r32["grad_colors"], r64["grad_colors"]))
    print(f"W{W} H{H} P{P} B{Bn}: image hip-vs-fp32 {e32:.1e} hip-vs-fp64 {e64:.1e} (fp32-vs-fp64 {r3264:.1e}) | grad colors hip-vs-fp32 {g32:.1e} hip-vs-fp64 {g64:.1e} (fp32-vs-fp64 {gr:.1e})", flush=True)
