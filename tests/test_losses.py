"""spectral / stencil losses.

CPU: the torch formulation (oracle/torch_losses.py, the checker) against the reference fixtures G11.
GPU: the product (fresnel_amd/losses.py -> fgs_spectral_loss_* / fgs_helmholtz_loss_* in libfgs_hip.so) against the
same fixtures (loss and every gradient <= 1e-4), and against the checker at a larger, non-square, batched size."""
import numpy as np
import pytest
import torch

from helpers import load_golden, rel_to_max

TOL = 1e-4


def _cases(mod, dev):
    P, Fq, helm = mod.PhaseRetrievalLoss, mod.FrequencyDomainLoss, mod.wave_equation_loss
    return {
        "phase": (lambda r, t, d: P(wavelength=0.05, focal_depth=0.5)(r, t, d), ("rendered", "target", "depth")),
        "phase_wl": (lambda r, t, d: P()(r, t, d.unsqueeze(1), wavelength=torch.tensor(0.0635, device=dev)),
                     ("rendered", "target", "depth")),
        "freq": (lambda r, t: Fq(cutoff=0.1, high_weight=2.0)(r, t), ("rendered", "target")),
        "freq_c25": (lambda r, t: Fq(cutoff=0.25, high_weight=0.5)(r, t), ("rendered", "target")),
        "helm": (lambda u: helm(u, 0.05), ("rendered",)),
        "helm3": (lambda u: helm(u, 0.0635, pixel_spacing=1.0 / 128.0), ("depth",)),
    }


def _run(mod, dev):
    g = load_golden("G11_losses_48x40")
    for tag, (fn, names) in _cases(mod, dev).items():
        ts = [torch.tensor(g[n], device=dev, requires_grad=True) for n in names]
        loss = fn(*ts)
        loss.backward()
        ref = float(g[tag + "_loss"])
        assert abs(loss.item() - ref) <= TOL * abs(ref), (tag, loss.item(), ref)
        for i, t in enumerate(ts):
            assert rel_to_max(t.grad.cpu().numpy(), g[f"{tag}_grad{i}"]) <= TOL, (tag, i)


def test_checker_matches_reference_fixtures_cpu():
    from oracle import torch_losses
    _run(torch_losses, torch.device("cpu"))


def test_product_losses_refuse_cpu_tensors():
    from fresnel_amd import _binding as B
    from fresnel_amd.losses import FrequencyDomainLoss, PhaseRetrievalLoss, wave_equation_loss
    x = torch.rand(1, 3, 8, 8)
    for fn in (lambda: FrequencyDomainLoss()(x, x), lambda: PhaseRetrievalLoss()(x, x, x[:, 0]),
               lambda: wave_equation_loss(x, 0.05)):
        with pytest.raises(B.FgsError):
            fn()


@pytest.mark.gpu
def test_hip_losses_match_reference_fixtures_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test requires an MI355X (torch.cuda unavailable)")
    from fresnel_amd import losses
    _run(losses, torch.device("cuda:0"))


@pytest.mark.gpu
def test_hip_losses_match_checker_at_render_size_gpu():
    """B = 3 images of 3 x 160 x 96 (non-square, not a power of two), learnable wavelength: loss and the gradients
    with respect to rendered, target, depth and the wavelength against the torch formulation run in float64."""
    from fresnel_amd import losses
    from oracle import torch_losses
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(3)
    Bn, H, W = 3, 160, 96
    rendered = rs.uniform(0, 1, (Bn, 3, H, W)).astype(np.float32)
    rendered[1, :, :7, :5] = 0.0
    target = rs.uniform(0, 1, (Bn, 3, H, W)).astype(np.float32)
    depth = rs.uniform(0.1, 2.5, (Bn, H, W)).astype(np.float32)

    def run(mod, dtype, device):
        r, t, d = [torch.tensor(a, dtype=dtype, device=device, requires_grad=True) for a in (rendered, target, depth)]
        wl = torch.tensor(0.0575, dtype=dtype, device=device, requires_grad=True)
        total = (mod.PhaseRetrievalLoss(focal_depth=0.7)(r, t, d, wavelength=wl) * 1.5 +
                 mod.FrequencyDomainLoss(cutoff=0.2, high_weight=3.0)(r, t) * 0.25 +
                 mod.wave_equation_loss(r, 0.05, pixel_spacing=1.0 / 96) * 1e-9)
        total.backward()
        return [float(total)] + [x.grad.detach().cpu().double().numpy() for x in (r, t, d, wl)]

    got = run(losses, torch.float32, dev)
    ref = run(torch_losses, torch.float64, torch.device("cpu"))
    assert abs(got[0] - ref[0]) <= TOL * abs(ref[0])
    for a, b, name in zip(got[1:], ref[1:], ["rendered", "target", "depth", "wavelength"]):
        assert rel_to_max(a, b) <= TOL, name
