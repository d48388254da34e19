"""The real train step on one GPU: decoder -> hand-off -> HIP rasterizer -> losses -> backward ->
clip -> fused AdamW, fresnel_amd.train.train_step at BASELINE config 2 / config 3 shapes (N = 37*37*K Gaussians from the
stand-in decoder: K = 6 -> 8214 @ 256^2 x 16 images, K = 24 -> 32856 @ 512^2 x 8 images).  Reports step ms (host clock
over `steps` steps, one sync at the end), the rasterizer's share from the library's stage timers (a second pass with
every stage bracketed), pairs/s, and the number of host syncs per step (torch sync debug mode).
usage: python scratch/profile/train_step_bench.py [steps] > gpurun_out/train_step.json"""
import json, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, '.')
from fresnel_amd import _binding as B
from fresnel_amd.dist import DPContext
from fresnel_amd.train import GraphedTrainStep, PatchGaussianDecoder, SyntheticDataset, TrainingConfig, default_renderer_factory, make_optimizer, train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
out = {}
for name, K, S, Bn in (("config2_shape", 6, 256, 16), ("config3_shape", 24, 512, 8)):
    cfg = TrainingConfig(batch_size=Bn, image_size=S, gaussians_per_patch=K, device='cuda:0', ssim_weight=0.0)
    torch.manual_seed(0)
    model = PatchGaussianDecoder(cfg.feature_dim, K, grid=cfg.feature_size).to(dev)
    renderer, camera = default_renderer_factory(cfg, dev)
    pairs = torch.zeros(1, dtype=torch.int64, device=dev)
    renderer.pair_counter = pairs
    opt = make_optimizer(model, cfg)
    dp = DPContext(device=dev)
    data = SyntheticDataset(4 * Bn, cfg)
    batches = [data.batch(list(range(i * Bn, (i + 1) * Bn)), dev) for i in range(4)]
    rng = np.random.RandomState(0)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.5:  # plan caches, allocator, and the GPU's sustained clocks (~100 ms of load)
        for i in range(5):
            train_step(model, renderer, camera, batches[i % 4], opt, cfg, dp, pose_rng=rng)
        torch.cuda.synchronize()
    # host syncs inside a step: torch's sync debug mode warns once per synchronising call
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        train_step(model, renderer, camera, batches[0], opt, cfg, dp, pose_rng=rng)
    torch.cuda.set_sync_debug_mode("default")
    syncs = len([x for x in w if "synchroniz" in str(x.message).lower()])
    torch.cuda.synchronize(); pairs.zero_()
    t0 = time.perf_counter()
    for i in range(steps):
        res = train_step(model, renderer, camera, batches[i % 4], opt, cfg, dp, pose_rng=rng)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    P = int(pairs.item()) / steps
    B.stage_timing_enable(True); B.stage_timing_read()
    for i in range(5):
        train_step(model, renderer, camera, batches[i % 4], opt, cfg, dp, pose_rng=rng)
    torch.cuda.synchronize()
    st = {k: v[0] / v[1] for k, v in B.stage_timing_read().items() if v[1]}
    B.stage_timing_enable(False)
    ras = sum(st.values())
    # the same step replayed from ONE captured HIP graph (TrainingConfig.hip_graph)
    gms = None
    try:
        cfg_g = TrainingConfig(batch_size=Bn, image_size=S, gaussians_per_patch=K, device='cuda:0', ssim_weight=0.0, hip_graph=True)
        opt_g = make_optimizer(model, cfg_g)
        renderer.pair_counter = None
        g = GraphedTrainStep(model, renderer, camera, opt_g, cfg_g, dp, batches[0])
        for i in range(3):
            g(batches[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            g(batches[i % 4])
        torch.cuda.synchronize()
        gms = (time.perf_counter() - t0) / steps * 1e3
    except Exception as e:  # report, do not hide
        gms = f"failed: {type(e).__name__}: {e}"
    out[name] = dict(gaussians=37 * 37 * K, resolution=S, images=Bn, step_ms=round(ms, 3), rasterizer_ms=round(ras, 3),
                     rasterizer_share=round(ras / ms, 3), step_ms_hip_graph=(round(gms, 3) if isinstance(gms, float) else gms), pairs_per_step=int(P), pairs_per_s=P / (ms * 1e-3),
                     host_syncs_per_step=syncs, stage_ms={k: round(v, 4) for k, v in st.items()},
                     decoder_params=sum(p.numel() for p in model.parameters()), loss=res.to_host())
print(json.dumps(out))
