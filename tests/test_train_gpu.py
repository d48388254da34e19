"""GPU end-to-end test of the training harness through the HIP rasterizer (config-1-like
plumbing shape: 128x128, a few hundred Gaussians per image)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_harness_trains_through_hip_renderer(tmp_path):
    from fresnel_amd.train import TrainingConfig, run_training
    assert torch.cuda.is_available()
    cfg = TrainingConfig(batch_size=4, epochs=2, lr=5e-3, image_size=128, feature_size=8, feature_dim=16,
                         gaussians_per_patch=4, device="cuda:0", steps_per_epoch=4, save_interval=1,
                         output_dir=str(tmp_path), log_interval=1000)
    model, hist = run_training(cfg, log=lambda *a: None)
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["total"])) for h in hist)
    assert hist[-1]["total"] < hist[0]["total"]  # it learns something
    ck = torch.load(tmp_path / "decoder_exp2_epoch1.pt")
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "losses", "config"}  # TGD:1304-1310
    for p in model.parameters():
        assert torch.isfinite(p).all()


def test_harness_wave_and_phase_renderers(tmp_path):
    """--use_wave_rendering (WaveFieldRenderer) and --use_phase_blending --use_fresnel_zones (config 4)
    train through the HIP kernels without NaNs."""
    from fresnel_amd.train import TrainingConfig, run_training
    for kw in (dict(use_wave_rendering=True), dict(use_phase_blending=True, use_fresnel_zones=True)):
        cfg = TrainingConfig(batch_size=2, epochs=1, lr=2e-3, image_size=64, feature_size=6, feature_dim=16,
                             gaussians_per_patch=4, device="cuda:0", steps_per_epoch=3, save_interval=100,
                             output_dir=str(tmp_path), log_interval=1000, **kw)
        model, hist = run_training(cfg, log=lambda *a: None)
        assert len(hist) == 1 and "total" in hist[0] and hist[0]["total"] == hist[0]["total"]
        for p in model.parameters():
            assert torch.isfinite(p).all()



def test_harness_spectral_losses_with_wave_renderer(tmp_path):
    """SURVEY 8f N1 + N2 inside the harness: WaveFieldRenderer with the phase-retrieval, frequency-domain and
    Helmholtz losses switched on (torch.fft on the GPU = rocFFT).  The epoch losses carry the three extra terms,
    everything stays finite."""
    from fresnel_amd.train import TrainingConfig, run_training
    cfg = TrainingConfig(batch_size=2, epochs=1, lr=2e-3, image_size=64, feature_size=6, feature_dim=16,
                         gaussians_per_patch=4, device="cuda:0", steps_per_epoch=3, save_interval=100,
                         output_dir=str(tmp_path), log_interval=1000, use_wave_rendering=True,
                         use_phase_retrieval_loss=True, use_frequency_loss=True, wave_equation_weight=1e-12)
    model, hist = run_training(cfg, log=lambda *a: None)
    assert len(hist) == 1 and {"phase_retrieval", "frequency", "wave_eq"} <= set(hist[0])
    assert all(v == v and abs(v) < float("inf") for v in hist[0].values())
    for p in model.parameters():
        assert torch.isfinite(p).all()


def test_importance_subsample_gather_matches_torch_indexing():
    """fgs_gather_forward / backward (one launch for all Gaussian tensors) == the reference's advanced indexing
    `output[k][:, indices]` (TGD:1175-1184) and autograd's scatter, bit for bit; (B,N) and (B,N,3) phases; draws are
    without replacement and follow the opacity importance."""
    from fresnel_amd.handoff import importance_subsample, importance_weights
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    Bn, N, K = 3, 500, 120
    for pc in (0, 1, 3):
        out = dict(positions=torch.randn(Bn, N, 3, generator=g), scales=torch.rand(Bn, N, 3, generator=g),
                   rotations=torch.randn(Bn, N, 4, generator=g), colors=torch.rand(Bn, N, 3, generator=g),
                   opacities=torch.rand(Bn, N, generator=g))
        if pc:
            out["phases"] = torch.rand(Bn, N, generator=g) if pc == 1 else torch.rand(Bn, N, 3, generator=g)
        a = {k: v.to(dev).requires_grad_(True) for k, v in out.items()}
        b = {k: v.to(dev).requires_grad_(True) for k, v in out.items()}
        sub, idx = importance_subsample(a, K, generator=torch.Generator(device=dev).manual_seed(1))
        assert idx.shape == (K,) and len(set(idx.tolist())) == K  # without replacement
        ref = {k: v[:, idx] for k, v in b.items()}
        ws = {k: torch.randn(v.shape, generator=g).to(dev) for k, v in ref.items()}
        sum((sub[k] * ws[k]).sum() for k in sub).backward()
        sum((ref[k] * ws[k]).sum() for k in ref).backward()
        for k in out:
            assert torch.equal(sub[k], ref[k]), k
            assert torch.equal(a[k].grad, b[k].grad), k
    same, none_idx = importance_subsample(a, None)
    assert same is a and none_idx is None and importance_subsample(a, N)[0] is a
    # the draw follows p ~ batch-mean opacity: near-zero-opacity Gaussians are (almost) never kept
    opa = torch.full((2, 1000), 1e-9, device=dev)
    opa[:, :300] = 0.9
    many = dict(a, opacities=opa, positions=torch.zeros(2, 1000, 3, device=dev), scales=torch.zeros(2, 1000, 3, device=dev),
                rotations=torch.zeros(2, 1000, 4, device=dev), colors=torch.zeros(2, 1000, 3, device=dev))
    many.pop("phases", None)
    _, idx = importance_subsample(many, 200)
    assert int((idx < 300).sum()) >= 198
    assert abs(float(importance_weights(opa).sum()) - 1.0) < 1e-5


def test_harness_hfts_flags_and_data_dir(tmp_path):
    """The hand-off flags of configs 1 and 4 run through the harness on the GPU: --stochastic_k (importance
    subsampling through fgs_gather), --progressive_schedule, --train_resolution, --use_edge_aware,
    --multi_pose_augmentation with --use_pose_encoding (one orbit camera per batch), and --data_dir with feature /
    depth caches written in the reference's on-disk formats."""
    import numpy as np
    from PIL import Image
    from fresnel_amd.handoff import HFTSConfig
    from fresnel_amd.train import TrainingConfig, run_training
    rs = np.random.RandomState(0)
    d = tmp_path / "images"
    (d / "features").mkdir(parents=True)
    for i in range(4):
        Image.fromarray(rs.randint(0, 256, (40, 40, 3)).astype(np.uint8)).save(d / f"im{i}.png")
        (rs.standard_normal((37, 37, 8)).astype(np.float32) * 0.5).tofile(d / "features" / f"im{i}_dinov2.bin")
        rs.uniform(0, 1, (32, 32)).astype(np.float32).tofile(d / "features" / f"im{i}_depth.bin")
    cfg = TrainingConfig(batch_size=2, epochs=4, lr=2e-3, image_size=64, feature_size=37, feature_dim=8,
                         gaussians_per_patch=4, device="cuda:0", save_interval=100, output_dir=str(tmp_path / "ck"),
                         log_interval=1000, data_dir=str(d), use_edge_aware=True, multi_pose_augmentation=True,
                         use_pose_encoding=True, frontal_prob=0.5)
    hfts = HFTSConfig(train_resolution=48, progressive_schedule=True, stochastic_k=700)
    seen = []
    model, hist = run_training(cfg, hfts=hfts, log=lambda *a: seen.append(" ".join(str(x) for x in a)))
    assert any("Found 4 images" in s for s in seen)
    assert len(hist) == 4 and all(h["total"] == h["total"] for h in hist)
    for p in model.parameters():
        assert torch.isfinite(p).all()
    assert any(p.grad is not None and p.grad.abs().sum() > 0 for p in model.edge_detector.parameters())


def test_train_step_makes_no_host_synchronisation(tmp_path):
    """<= 1 host sync per step -- in fact none: loss terms stay on the device, the NaN/Inf skip flag
    rides in the gradient bucket and reaches the fused AdamW step as `found_inf`.  torch's sync debug mode turns any
    synchronising call (.item(), .tolist(), bool(tensor), a pageable H2D copy) into an error."""
    import numpy as np
    from fresnel_amd.dist import DPContext
    from fresnel_amd.train import (PatchGaussianDecoder, StepResult, SyntheticDataset, TrainingConfig,
                                   default_renderer_factory, make_optimizer, train_step)
    dev = torch.device("cuda:0")
    cfg = TrainingConfig(batch_size=2, epochs=1, lr=1e-3, image_size=64, feature_size=6, feature_dim=16, gaussians_per_patch=4,
                         device="cuda:0", use_frequency_loss=True, wave_equation_weight=1e-9)
    torch.manual_seed(0)
    model = PatchGaussianDecoder(cfg.feature_dim, cfg.gaussians_per_patch, grid=cfg.feature_size).to(dev)
    renderer, camera = default_renderer_factory(cfg, dev)
    opt = make_optimizer(model, cfg)
    assert opt.defaults.get("fused"), "the fused AdamW is what takes the skip flag on the device"
    dp = DPContext(device=dev)
    data = SyntheticDataset(8, cfg)
    rng = np.random.RandomState(0)
    batches = [data.batch([2 * i, 2 * i + 1], dev) for i in range(3)]
    train_step(model, renderer, camera, batches[0], opt, cfg, dp, pose_rng=rng)  # warm-up: plan caches, camera upload
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in model.parameters()]
    torch.cuda.set_sync_debug_mode("error")
    try:
        res = train_step(model, renderer, camera, batches[1], opt, cfg, dp, pose_rng=rng)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert isinstance(res, StepResult)
    ld = res.to_host()  # the one transfer, outside the step
    assert ld is not None and {"rgb", "depth", "frequency", "wave_eq", "total"} <= set(ld) and all(v == v for v in ld.values())
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    # a poisoned batch: skipped on the device -- weights and the optimizer's step count stay exactly as they were
    before = [p.detach().clone() for p in model.parameters()]
    steps = [float(opt.state[p]["step"]) for p in model.parameters()]
    bad = (batches[2][0] * float("nan"), batches[2][1], batches[2][2])
    res = train_step(model, renderer, camera, bad, opt, cfg, dp, pose_rng=rng)
    assert res.to_host() is None
    assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
    assert steps == [float(opt.state[p]["step"]) for p in model.parameters()]


def test_harness_writes_history_with_pairs_per_second_and_stage_ms(tmp_path):
    """SURVEY section 5 metrics: training_history_exp{E}.json carries the epoch losses (TGD:1317-1323) plus step ms,
    composited Gaussian-pixels/s (counted on the device by fgs_count_pairs) and the rasterizer's stage timers."""
    import json
    from fresnel_amd.train import TrainingConfig, run_training
    cfg = TrainingConfig(batch_size=2, epochs=2, lr=1e-3, image_size=64, feature_size=6, feature_dim=16, gaussians_per_patch=4,
                         device="cuda:0", steps_per_epoch=3, save_interval=100, output_dir=str(tmp_path), log_interval=2)
    seen = []
    run_training(cfg, log=lambda *a: seen.append(" ".join(str(x) for x in a)))
    h = json.load(open(tmp_path / "training_history_exp2.json"))
    assert len(h["total"]) == 2 and len(h["pairs_per_s"]) == 2 and all(v > 0 for v in h["pairs_per_s"]) and all(v > 0 for v in h["hbm_gbs_algorithmic"])
    assert {"composite_fwd", "composite_bwd", "project"} <= set(h["stage_ms"]) and h["skipped_batches"] == [0, 0]
    assert any(s.strip().startswith("Batch 0/3") for s in seen) and any("Batch 2/3" in s for s in seen)


def test_spectral_loss_refuses_a_second_backward():
    """fgs_spectral_loss_backward consumes the saved spectra in place; a second backward through the same node
    must raise instead of returning garbage."""
    from fresnel_amd.losses import FrequencyDomainLoss
    dev = torch.device("cuda:0")
    r = torch.rand(1, 3, 32, 32, device=dev, requires_grad=True)
    t = torch.rand(1, 3, 32, 32, device=dev)
    loss = FrequencyDomainLoss()(r, t)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()


def test_graph_captured_step_trains_like_the_eager_step(tmp_path):
    """--hip_graph: the whole step replayed from one captured HIP graph (the library neither allocates nor synchronises,
    the NaN/Inf skip is a device flag, the learning rate a device tensor the cosine schedule fills in place) must train
    exactly like the eager step: same losses, same parameters up to fp32 reduction order."""
    from fresnel_amd.train import TrainingConfig, run_training

    def run(graph):
        cfg = TrainingConfig(batch_size=2, epochs=3, lr=2e-3, image_size=64, feature_size=6, feature_dim=16, gaussians_per_patch=4,
                             device="cuda:0", steps_per_epoch=4, save_interval=100, output_dir=str(tmp_path / f"g{int(graph)}"),
                             log_interval=1000, hip_graph=graph, use_frequency_loss=True)
        return run_training(cfg, log=lambda *a: None)
    m_e, h_e = run(False)
    m_g, h_g = run(True)
    assert len(h_g) == 3
    for a, b in zip(h_e, h_g):
        assert set(a) == set(b)
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(a[k])), (k, a[k], b[k])
    for p, q in zip(m_e.parameters(), m_g.parameters()):
        assert torch.allclose(p, q, rtol=1e-3, atol=1e-5)
    assert h_g[-1]["total"] < h_g[0]["total"]  # the cosine schedule's lr reached the replayed graph and it learns
