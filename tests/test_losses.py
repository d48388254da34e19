"""spectral / stencil losses vs the reference fixtures (G11), on CPU and on the GPU."""
import numpy as np
import pytest
import torch

from helpers import load_golden, rel_to_max

TOL = 1e-4


def _run(dev):
    from fresnel_amd.losses import PhaseRetrievalLoss, FrequencyDomainLoss, wave_equation_loss
    g = load_golden("G11_losses_48x40")
    cases = {
        "phase": (lambda r, t, d: PhaseRetrievalLoss(wavelength=0.05, focal_depth=0.5)(r, t, d), ("rendered", "target", "depth")),
        "phase_wl": (lambda r, t, d: PhaseRetrievalLoss()(r, t, d.unsqueeze(1), wavelength=torch.tensor(0.0635, device=dev)),
                     ("rendered", "target", "depth")),
        "freq": (lambda r, t: FrequencyDomainLoss(cutoff=0.1, high_weight=2.0)(r, t), ("rendered", "target")),
        "freq_c25": (lambda r, t: FrequencyDomainLoss(cutoff=0.25, high_weight=0.5)(r, t), ("rendered", "target")),
        "helm": (lambda u: wave_equation_loss(u, 0.05), ("rendered",)),
        "helm3": (lambda u: wave_equation_loss(u, 0.0635, pixel_spacing=1.0 / 128.0), ("depth",)),
    }
    for tag, (fn, names) in cases.items():
        ts = [torch.tensor(g[n], device=dev, requires_grad=True) for n in names]
        loss = fn(*ts)
        loss.backward()
        ref = float(g[tag + "_loss"])
        assert abs(loss.item() - ref) <= TOL * abs(ref), (tag, loss.item(), ref)
        for i, t in enumerate(ts):
            assert rel_to_max(t.grad.cpu().numpy(), g[f"{tag}_grad{i}"]) <= TOL, (tag, i)


def test_losses_match_reference_fixtures_cpu():
    _run(torch.device("cpu"))


def test_frequency_weight_is_cached_and_masks_partition_the_spectrum():
    from fresnel_amd.losses import FrequencyDomainLoss
    fl = FrequencyDomainLoss(cutoff=0.1, high_weight=3.0)
    w = fl._weight(16, 12, torch.device("cpu"))
    assert w is fl._weight(16, 12, torch.device("cpu"))
    assert set(np.unique(w.numpy()).tolist()) == {1.0, 3.0} and w[0, 0] == 1.0


@pytest.mark.gpu
def test_losses_match_reference_fixtures_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test requires an MI355X (torch.cuda unavailable)")
    _run(torch.device("cuda:0"))
