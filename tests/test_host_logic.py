"""CPU tests of host-side logic that mirrors reference interfaces: 14-float binary format
(DR:1461-1497), orbit camera (TGD:684-757), Camera defaults (DR:27-52), decoder output
shapes/ranges, training flags."""
import numpy as np
import pytest
import torch

from helpers import load_golden


def test_binary_roundtrip_and_layout(tmp_path):
    from fresnel_amd.io import load_gaussians_from_binary, save_gaussians_to_binary
    g = {"positions": torch.randn(7, 3), "scales": torch.rand(7, 3), "rotations": torch.randn(7, 4),
         "colors": torch.rand(7, 3), "opacities": torch.rand(7)}
    p = str(tmp_path / "g.bin")
    save_gaussians_to_binary(p, g)
    raw = np.fromfile(p, dtype=np.float32).reshape(7, 14)
    assert np.array_equal(raw[:, 6:10], g["rotations"].numpy()) and np.array_equal(raw[:, 13], g["opacities"].numpy())
    back = load_gaussians_from_binary(p)
    for k in g:
        assert torch.equal(back[k], g[k])


def test_orbit_camera_matches_reference_view_matrix():
    """G7's view matrix was produced with the reference's formula at el=20, az=135 degrees."""
    from fresnel_amd.renderer import create_camera_from_pose
    g = load_golden("G7_orbit256_96")
    cam = create_camera_from_pose(np.deg2rad(20.0), np.deg2rad(135.0), 96)
    assert np.allclose(cam.view_matrix.numpy(), g["view"], atol=1e-6)
    assert cam.fx == 96 * 0.8 and cam.cx == 48 and cam.near == 0.01 and cam.far == 100.0


def test_camera_project_and_packing():
    from fresnel_amd.renderer import Camera
    cam = Camera(80.0, 80.0, 50.0, 40.0, 100, 80)
    assert torch.equal(cam.view_matrix, torch.eye(4))
    uv, d = cam.project(torch.tensor([[0.0, 0.0, -2.0], [0.5, 0.25, -4.0]]))
    assert torch.allclose(uv[0], torch.tensor([50.0, 40.0])) and torch.allclose(d, torch.tensor([2.0, 4.0]))
    assert torch.allclose(uv[1], torch.tensor([50.0 + 80 * 0.5 / 4, 40.0 - 80 * 0.25 / 4]))
    rec = cam.packed()
    assert len(rec) == 24 and rec[16:22] == [80.0, 80.0, 50.0, 40.0, 0.01, 100.0]


def test_decoder_output_contract():
    from fresnel_amd.decoder import PatchGaussianDecoder
    m = PatchGaussianDecoder(feature_dim=384, gaussians_per_patch=4, use_fresnel_zones=True, use_phase_output=True)
    out = m(torch.randn(2, 37, 37, 384), torch.rand(2, 1, 64, 64))
    assert out["positions"].shape == (2, 5476, 3) and out["phases"].shape == (2, 5476)
    assert out["scales"].min() >= 1e-6 and out["scales"].max() <= 2.0
    assert torch.allclose(out["rotations"].norm(dim=-1), torch.ones(2, 5476), atol=1e-5)
    for k in ("colors", "opacities", "phases"):
        assert out[k].min() >= 0 and out[k].max() <= 1
    assert len(torch.unique(out["positions"][..., 2])) <= 8  # zone-snapped depths (config 4)
    n_params = sum(p.numel() for p in m.parameters())
    assert 0.5e6 < n_params < 0.8e6


def test_train_cli_rejects_cpu_and_other_experiments():
    import pytest
    from fresnel_amd import train
    with pytest.raises(SystemExit):
        train.main(["--experiment", "3"])
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit):
            train.main(["--experiment", "2", "--epochs", "1"])


def test_compute_losses_adds_the_spectral_terms_with_the_reference_weights(monkeypatch):
    """TGD:957-996: wave-equation, phase-retrieval and frequency losses join the total with their weights.  (Host
    logic only: the product's loss kernels need a GPU, so the checker's torch formulation stands in for them here.)"""
    import torch
    from fresnel_amd import losses as product_losses
    from fresnel_amd import train as T
    from oracle.torch_losses import FrequencyDomainLoss, PhaseRetrievalLoss, wave_equation_loss
    for name, obj in (("FrequencyDomainLoss", FrequencyDomainLoss), ("PhaseRetrievalLoss", PhaseRetrievalLoss),
                      ("wave_equation_loss", wave_equation_loss)):
        monkeypatch.setattr(product_losses, name, obj)
    monkeypatch.setattr(T, "_LOSS_MODULES", {})  # fresh cache for this test only
    g = torch.Generator().manual_seed(3)
    r, t = torch.rand(2, 3, 16, 16, generator=g), torch.rand(2, 3, 16, 16, generator=g)
    rd, td = torch.rand(2, 16, 16, generator=g), torch.rand(2, 16, 16, generator=g)
    base_cfg = T.TrainingConfig(image_size=16, ssim_weight=0.0)
    base, _ = T.compute_losses(r, t, rd, td, base_cfg)
    cfg = T.TrainingConfig(image_size=16, ssim_weight=0.0, wave_equation_weight=1e-9, use_phase_retrieval_loss=True,
                           use_frequency_loss=True)
    total, d = T.compute_losses(r, t, rd, td, cfg)
    want = (base + 1e-9 * wave_equation_loss(r, 0.05, pixel_spacing=1.0 / 16) + 0.1 * PhaseRetrievalLoss()(r, t, td) +
            0.1 * FrequencyDomainLoss()(r, t))
    assert abs(float(total) - float(want)) <= 1e-5 * abs(float(want))
    assert {"wave_eq", "phase_retrieval", "frequency"} <= set(d)


def test_column_fft_pass_structure_numpy_model():
    """The in-LDS column FFT of fgs_colfft.h restated index for index in numpy and checked against numpy.fft: radix-8
    passes (three radix-2 stages on the points i + k M/8, twiddles w_M^i w_8^k / w_M^2i w_4^k' / w_M^4i), then one radix-4
    or radix-2 pass for the remaining stages; decimation in frequency forward -> bit-reversed order out, decimation in
    time inverse <- bit-reversed order in.  The kernels themselves are checked on the GPU (tests/test_hip_asm.py,
    tests/test_losses.py); this pins the scheme."""
    S = np.sqrt(0.5)
    W8 = [1, S * (1 - 1j), -1j, S * (-1 - 1j)]

    def bitrev(r, logn):
        return int(format(r, "0%db" % logn)[::-1], 2)

    def fft_cols(x, logn, inv):
        N = 1 << logn
        x = x.astype(np.complex128).copy()
        tw = np.exp(-2j * np.pi * np.arange(N // 2) / N)

        def pair():
            for k in range(N // 2):
                a, b = x[2 * k], x[2 * k + 1]
                x[2 * k], x[2 * k + 1] = a + b, a - b

        def quad(M):
            Q, step = M // 4, N // M
            for q in range(N // 4):
                i = q % Q
                p0 = (q // Q) * M + i
                p1, p2, p3 = p0 + Q, p0 + 2 * Q, p0 + 3 * Q
                w1, w2 = tw[i * step], tw[2 * i * step]
                a0, a1, a2, a3 = x[p0], x[p1], x[p2], x[p3]
                if not inv:
                    s02, s13, d02, d13 = a0 + a2, a1 + a3, a0 - a2, a1 - a3
                    u2, u3 = d02 * w1, (d13 * -1j) * w1
                    x[p0], x[p1], x[p2], x[p3] = s02 + s13, (s02 - s13) * w2, u2 + u3, (u2 - u3) * w2
                else:
                    t1, t3 = a1 * np.conj(w2), a3 * np.conj(w2)
                    r0, r1, r2, r3 = a0 + t1, a0 - t1, a2 + t3, a2 - t3
                    v2, v3 = r2 * np.conj(w1), (r3 * np.conj(w1)) * 1j
                    x[p0], x[p2], x[p1], x[p3] = r0 + v2, r0 - v2, r1 + v3, r1 - v3

        def octp(M):
            E, step = M // 8, N // M
            for q in range(N // 8):
                i = q % E
                p = [(q // E) * M + i + k * E for k in range(8)]
                t1, t2, t4 = tw[i * step], tw[2 * i * step], tw[4 * i * step]
                a = [x[pp] for pp in p]
                if not inv:
                    s = [a[k] + a[k + 4] for k in range(4)]
                    d = [(a[k] - a[k + 4]) * W8[k] * t1 for k in range(4)]
                    b = []
                    for h in (s, d):
                        b += [h[0] + h[2], h[1] + h[3], (h[0] - h[2]) * t2, (h[1] - h[3]) * (-1j) * t2]
                    out = []
                    for qd in range(4):
                        out += [b[2 * qd] + b[2 * qd + 1], (b[2 * qd] - b[2 * qd + 1]) * t4]
                else:
                    c1, c2, c4 = np.conj(t1), np.conj(t2), np.conj(t4)
                    b = []
                    for qd in range(4):
                        v = a[2 * qd + 1] * c4
                        b += [a[2 * qd] + v, a[2 * qd] - v]
                    sd = []
                    for h in (b[0:4], b[4:8]):
                        v2, v3 = h[2] * c2, h[3] * c2 * 1j
                        sd.append([h[0] + v2, h[1] + v3, h[0] - v2, h[1] - v3])
                    out = [0] * 8
                    for k in range(4):
                        v = sd[1][k] * np.conj(W8[k]) * c1
                        out[k], out[k + 4] = sd[0][k] + v, sd[0][k] - v
                for k in range(8):
                    x[p[k]] = out[k]

        seq, lg = [], logn
        while lg >= 3:
            seq.append(("o", lg))
            lg -= 3
        if lg == 2:
            seq.append(("q", 2))
        if lg == 1:
            seq.append(("p", 1))
        for kind, l in (seq[::-1] if inv else seq):
            octp(1 << l) if kind == "o" else (quad(1 << l) if kind == "q" else pair())
        return x

    rs = np.random.RandomState(0)
    for logn in (6, 7, 8, 9, 10):
        N = 1 << logn
        br = np.array([bitrev(r, logn) for r in range(N)])
        v = rs.randn(N) + 1j * rs.randn(N)
        assert np.abs(fft_cols(v, logn, False) - np.fft.fft(v)[br]).max() < 1e-10 * N
        g = rs.randn(N) + 1j * rs.randn(N)
        assert np.abs(fft_cols(g[br], logn, True) - np.fft.ifft(g) * N).max() < 1e-10 * N


def test_epoch_batches_shuffle_per_epoch_same_on_all_ranks_and_keep_the_tail():
    """a fresh permutation per epoch from seed + epoch (what every rank draws), the last partial
    batch kept when it divides over the ranks (DataLoader(shuffle=True) without drop_last, TGD:1760-1767)."""
    from fresnel_amd.train import epoch_batches
    e0, e1 = epoch_batches(22, 4, 2, seed=5, epoch=0), epoch_batches(22, 4, 2, seed=5, epoch=1)
    assert e0 == epoch_batches(22, 4, 2, seed=5, epoch=0)          # deterministic: identical on every rank
    assert e0 != e1                                                # reshuffled every epoch
    assert sorted(i for b in e0 for i in b) == list(range(22))     # tail of 2 kept: divides over 2 ranks
    assert [len(b) for b in e0] == [4] * 5 + [2]
    assert sum(len(b) for b in epoch_batches(23, 4, 2, seed=5, epoch=0)) == 20   # tail of 3 on 2 ranks: dropped
    assert e0 != [list(range(i, i + 4)) for i in range(0, 20, 4)] + [[20, 21]]   # not the sorted-filename order


def test_batch_prefetcher_yields_the_batches_in_order_and_surfaces_errors():
    import torch
    from fresnel_amd.train import BatchPrefetcher, SyntheticDataset, TrainingConfig
    cfg = TrainingConfig(image_size=16, feature_size=4, feature_dim=8, device="cpu")
    data = SyntheticDataset(12, cfg)
    lists = [[3, 1], [0, 7], [11, 2], [5]]
    got = list(BatchPrefetcher(data, lists, "cpu", num_workers=3, depth=2))
    assert len(got) == 4
    for idx, b in zip(lists, got):
        want = data.batch(idx, "cpu")
        assert all(torch.equal(x, y) for x, y in zip(b, want))

    class Broken(SyntheticDataset):
        def host_item(self, i):
            raise OSError("unreadable image")
    with pytest.raises(OSError):
        list(BatchPrefetcher(Broken(4, cfg), [[0, 1]], "cpu"))


def test_batch_prefetcher_producer_ends_when_the_consumer_leaves_early():
    """a consumer that stops early (break / an exception in the step) must not leave the producer thread
    blocked in `put` holding pinned batches."""
    from fresnel_amd.train import BatchPrefetcher, SyntheticDataset, TrainingConfig
    cfg = TrainingConfig(image_size=16, feature_size=4, feature_dim=8, device="cpu")
    data = SyntheticDataset(64, cfg)
    pf = BatchPrefetcher(data, [[i, i + 1] for i in range(0, 60, 2)], "cpu", num_workers=2, depth=1)
    for k, _ in enumerate(pf):
        if k == 1:
            break           # generator closed -> finally: close()
    pf.thread.join(timeout=10.0)
    assert not pf.thread.is_alive() and pf.q.empty()
    pf2 = BatchPrefetcher(data, [[i, i + 1] for i in range(0, 60, 2)], "cpu", num_workers=2, depth=1)
    with pytest.raises(RuntimeError):
        for _ in pf2:
            raise RuntimeError("train_step failed")
    pf2.thread.join(timeout=10.0)
    assert not pf2.thread.is_alive()


def test_image_dataset_host_item_does_not_decode_the_saag_binaries(tmp_path, monkeypatch):
    """The prefetch threads load image, features and depth only (the step never uses the SAAG Gaussians)."""
    import numpy as np
    from PIL import Image
    from fresnel_amd import data as D, io as fio
    Image.fromarray((np.random.RandomState(0).rand(20, 20, 3) * 255).astype(np.uint8)).save(tmp_path / "a.png")
    (tmp_path / "features").mkdir()
    (tmp_path / "features" / "a_saag.bin").write_bytes(b"\0" * 64)
    monkeypatch.setattr(fio, "load_gaussians_from_binary", lambda *_: (_ for _ in ()).throw(AssertionError("decoded SAAG")))
    ds = D.ImageDataset(str(tmp_path), image_size=16)
    img, feats, dep = ds.host_item(0)
    assert img.shape == (3, 16, 16) and feats.shape == (37, 37, 384) and dep.shape == (1, 16, 16)


def test_harness_selects_the_zone_key_depth_sort_with_fresnel_zones():
    """--use_fresnel_zones snaps the decoder's depths to a few values (GDM:834-841): the harness then asks the renderer for the
    zone-key depth sort (FgsDims.sort_mode = 1, one radix pass instead of four); without the flag nothing is overridden."""
    from fresnel_amd.train import TrainingConfig, default_renderer_factory
    ren, cam = default_renderer_factory(TrainingConfig(image_size=64, device="cpu", use_fresnel_zones=True), "cpu")
    assert ren.tuning == dict(sort_mode=1) and cam.width == 64
    ren, _ = default_renderer_factory(TrainingConfig(image_size=64, device="cpu"), "cpu")
    assert ren.tuning is None
    from fresnel_amd import _binding as B
    d = B.make_dims(2, 100, 64, 64, tuning=dict(sort_mode=1))
    assert d.sort_mode == 1 and B.make_dims(2, 100, 64, 64).sort_mode == 0
    with pytest.raises(B.FgsError):
        B.saved_layout(B.make_dims(2, 100, 64, 64, tuning=dict(sort_mode=12)))
    with pytest.raises(B.FgsError):
        B.saved_layout(B.make_dims(70000, 4, 16, 16))   # images are a grid dimension: <= 65535


def test_wave_renderers_follow_their_background_buffer():
    """the kernels take the background as host floats; the module must notice when the registered buffer is
    edited in place, replaced, or loaded from a state dict (the reference reads the buffer on every call), and must NOT
    read it back when nothing changed or after a plain device / dtype move."""
    import torch
    from fresnel_amd.renderer import ASMWaveFieldRenderer, WaveFieldRenderer
    for cls in (ASMWaveFieldRenderer, WaveFieldRenderer):
        r = cls(32, 32, background=(0.1, 0.2, 0.3))
        assert r._background_host() == pytest.approx([0.1, 0.2, 0.3])
        marker = r._bg
        assert r._background_host() is marker                       # unchanged buffer: no read-back
        r = r.to(torch.float32)
        assert r._background_host() is marker                       # a move keeps the values: no read-back
        r.background.mul_(2.0)
        assert r._background_host() == pytest.approx([0.2, 0.4, 0.6])
        sd = r.state_dict()
        sd["background"] = torch.tensor([0.5, 0.25, 0.0])
        r.load_state_dict(sd)
        assert r._background_host() == pytest.approx([0.5, 0.25, 0.0])
        r.background = torch.tensor([1.0, 0.0, 0.0])
        assert r._background_host() == [1.0, 0.0, 0.0]


def test_training_history_json_and_device_side_nan_skip(tmp_path):
    """run_training on CPU with a stand-in renderer: history file of TGD:1317-1323 (+ step_ms), and a poisoned batch
    is skipped by the fused optimizer's found_inf path without touching weights or step count."""
    import json
    import torch
    from fresnel_amd import train as T

    class Ren(torch.nn.Module):
        def __init__(self, res, poison):
            super().__init__()
            self.res, self.poison, self.calls = res, poison, 0

        def forward(self, pos, scale, rot, col, opa, camera, return_depth=False, phases=None):
            self.calls += 1
            img = (col.mean(1)[:, :, None, None] + 0 * (pos.sum() + scale.sum() + rot.sum() + opa.sum())).expand(-1, 3, self.res, self.res)
            if self.calls in self.poison:
                img = img * float("nan")
            return img, (-pos[..., 2]).mean(1)[:, None, None].expand(-1, self.res, self.res) + 0 * opa.sum()

    def run(poison):
        cfg = T.TrainingConfig(batch_size=2, epochs=2, lr=1e-2, image_size=8, feature_size=3, feature_dim=4, gaussians_per_patch=1,
                               ssim_weight=0.0, device="cpu", steps_per_epoch=3, save_interval=100, log_interval=100,
                               output_dir=str(tmp_path / f"o{len(poison)}"), num_workers=1)
        ren = Ren(8, poison)
        model, hist = T.run_training(cfg, renderer_factory=lambda c, d, r: (ren, None), log=lambda *a: None)
        return cfg, model, hist
    cfg, model, hist = run(set())
    h = json.load(open(tmp_path / "o0" / "training_history_exp2.json"))
    assert {"total", "rgb", "depth", "step_ms", "skipped_batches"} <= set(h) and len(h["total"]) == 2
    assert h["skipped_batches"] == [0, 0] and abs(h["total"][-1] - hist[-1]["total"]) < 1e-12
    # every batch of the run poisoned: nothing learned, parameters exactly the initial ones, history has no losses
    cfg2, model2, hist2 = run(set(range(1, 100)))
    torch.manual_seed(cfg2.seed)
    fresh = T.PatchGaussianDecoder(cfg2.feature_dim, cfg2.gaussians_per_patch, grid=cfg2.feature_size)
    for a, b in zip(model2.parameters(), fresh.parameters()):
        assert torch.equal(a, b)
    assert hist2 == [{}, {}]
    # one poisoned batch: skipped, the others train
    _, model3, hist3 = run({2})
    h3 = json.load(open(tmp_path / "o1" / "training_history_exp2.json"))
    assert h3["skipped_batches"] == [1, 0] and all(torch.isfinite(p).all() for p in model3.parameters())


def test_camera_record_cache_follows_set_view_and_data_edits():
    """the packed-camera cache must not serve a stale record after set_view (id() reuse) or an edit through
    .data (which leaves _version unchanged)."""
    import torch
    from fresnel_amd.renderer import Camera
    cam = Camera(10.0, 10.0, 4.0, 4.0, 8, 8)
    r0 = cam.packed_tensor("cpu").clone()
    for k in range(3):  # fresh matrices that may reuse a freed tensor's id with _version 0
        v = torch.eye(4)
        v[0, 3] = float(k + 1)
        cam.set_view(v)
        assert cam.packed_tensor("cpu")[0, 3].item() == float(k + 1)
        del v
    cam.view_matrix.data[1, 3] = 7.0
    assert cam.packed_tensor("cpu")[0, 7].item() == 7.0
    assert cam.packed_tensor("cpu") is cam.packed_tensor("cpu")  # unchanged camera: one upload
    assert not torch.equal(r0, cam.packed_tensor("cpu"))


def test_phase_tensor_shapes_are_validated_before_any_kernel_sees_them():
    import torch
    from fresnel_amd.renderer import _phase_channels
    assert _phase_channels(torch.zeros(2, 5)) == 1 and _phase_channels(torch.zeros(2, 5, 3)) == 3
    assert _phase_channels(torch.zeros(2, 5, 1)) == 1   # one float per Gaussian, NOT three
    with pytest.raises(ValueError):
        _phase_channels(torch.zeros(2, 5, 2))


def test_plane_recurrence_identities_of_the_asm_column_kernels():
    """The algebra k_colfft_fwd / k_colfft_bwd (fgs_asm.hip) rest on since round 3, restated in numpy: equally spaced planes give
    H_p = H_lo D^(p - lo), hence  sum_p H_p F_p = H_lo S  and  sum_p z_p H_p F_p = H_lo (z_lo S - step T)  with the Horner sums
    S_k = F_k + D S_(k+1), T_k = D (T_(k+1) + S_(k+1)) over the planes in descending order -- planes without Gaussians (F = 0) are
    stepped over with the same recurrence -- and the backward's gF_p = g conj(H_p) is w <- w conj(D) from w = g conj(H_lo)."""
    rs = np.random.RandomState(5)
    for P, lo, hi, occupied in [(16, 0, 16, None), (16, 6, 12, [7, 8, 11]), (6, 4, 6, [4]), (64, 0, 64, [3, 40, 60, 63]), (5, 0, 5, [])]:
        n = 37
        near, far, focal = 0.3, 2.4, 0.9
        step = (far - near) / (P - 1)
        z = focal - (near + step * np.arange(P))                       # z_p, equally spaced
        kz = rs.uniform(0.0, 23.0, n)
        H = np.exp(1j * 2 * np.pi * z[:, None] * kz[None, :])          # H_p as the table holds it
        D = np.exp(1j * 2 * np.pi * (-step) * kz)
        occ = list(range(lo, hi)) if occupied is None else occupied
        F = np.zeros((P, n), complex)
        for p in occ:
            F[p] = rs.standard_normal(n) + 1j * rs.standard_normal(n)
        acc = sum(H[p] * F[p] for p in range(lo, hi))
        Z = sum(z[p] * H[p] * F[p] for p in range(lo, hi))
        S = np.zeros(n, complex); T = np.zeros(n, complex)
        prev = None
        for p in sorted(occ, reverse=True):                            # the kernel's walk: occupied planes, descending
            for _ in range(0 if prev is None else prev - p - 1):       # skip_planes: F = 0
                T = D * (T + S); S = D * S
            T = D * (T + S); S = F[p] + D * S                           # plane_step's epilogue
            prev = p
        for _ in range(0 if prev is None else prev - lo):               # down to the group's first plane
            T = D * (T + S); S = D * S
        assert np.abs(H[lo] * S - acc).max() <= 1e-9 * max(1.0, np.abs(acc).max())
        assert np.abs(H[lo] * (z[lo] * S - step * T) - Z).max() <= 1e-9 * max(1.0, np.abs(Z).max())
        g = rs.standard_normal(n) + 1j * rs.standard_normal(n)
        w = g * np.conj(H[lo])
        for p in range(lo, hi):
            assert np.abs(w - g * np.conj(H[p])).max() <= 1e-9
            w = w * np.conj(D)


def test_renderers_tolerate_inference_mode_tensors():
    """tensors created under torch.inference_mode() have no version counter (`_version` raises).  The wave renderers'
    host copy of the background buffer and the camera's packed-record cache are keyed by version: constructing / moving the modules
    or packing a camera inside inference mode must work (the cache then counts as stale: one read-back, never an exception)."""
    from fresnel_amd.renderer import ASMWaveFieldRenderer, Camera, WaveFieldRenderer, _tensor_version
    with torch.inference_mode():
        r = ASMWaveFieldRenderer(32, 32, background=(0.1, 0.2, 0.3)).to("cpu")
        w = WaveFieldRenderer(16, 16, background=(0.0, 0.5, 1.0))
        assert _tensor_version(r.background) is None
        assert r._background_host() == pytest.approx([0.1, 0.2, 0.3]) and w._background_host() == pytest.approx([0.0, 0.5, 1.0])
        cam = Camera(10.0, 10.0, 8.0, 8.0, 16, 16)
        cam.set_view(torch.eye(4))
        assert cam.packed_tensor("cpu").shape == (1, 24)
    assert r._background_host() == pytest.approx([0.1, 0.2, 0.3])  # ... and outside it afterwards
    n = WaveFieldRenderer(16, 16, background=(0.2, 0.2, 0.2))       # a normal module still notices an in-place edit
    n.background.mul_(2.0)
    assert n._background_host() == pytest.approx([0.4, 0.4, 0.4])


def test_call_shape_cache_and_input_normalisation():
    """Round 5 (the per-image drop-in route makes B small calls per step): FgsDims and both workspace sizes are built once per
    distinct call shape / configuration; inputs that already are contiguous fp32 pass through without a copy."""
    from fresnel_amd import renderer as R
    cfg = R._Cfg(64, 48, (0.1, 0.2, 0.3), 64, False, 0.25)
    a = R._dims_for(2, 100, cfg, False, 1)
    assert R._dims_for(2, 100, cfg, False, 1) is a and R._dims_for(3, 100, cfg, False, 1) is not a
    assert a[0].batch == 2 and a[0].num_gaussians == 100 and a[1] > 0 and a[2] > 0
    from fresnel_amd import _binding as B
    assert (a[1], a[2]) == B.workspace_bytes(a[0])
    cfg2 = R._Cfg(64, 48, (0.1, 0.2, 0.3), 64, False, 0.25, tuning=dict(tile_w=16))
    assert R._dims_for(2, 100, cfg2, False, 1) is not a and R._dims_for(2, 100, cfg2, False, 1)[0].tile_w == 16
    t = torch.zeros(4, 3)
    assert R._f32c(t) is t
    g = torch.zeros(4, 3, requires_grad=True)
    assert R._f32c(g).data_ptr() == g.data_ptr() and not R._f32c(g).requires_grad
    h = torch.zeros(3, 4, dtype=torch.float64).t()
    assert R._f32c(h).dtype == torch.float32 and R._f32c(h).is_contiguous()
