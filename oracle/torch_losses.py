"""Torch formulation of the spectral / stencil losses -- TEST INFRASTRUCTURE (CPU checker): only tests/ import it.

Restates PhaseRetrievalLoss (TGD:342-425), FrequencyDomainLoss (TGD:428-522) and wave_equation_loss (TGD:781-835)
with stock torch ops; pinned by the reference fixtures G11 (tests/test_losses.py) and used as the reference the HIP
implementation (fresnel_amd/losses.py -> libfgs_hip.so) is compared with on the GPU at sizes beyond the fixtures.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import nn


class PhaseRetrievalLoss(nn.Module):
    """MSE between the magnitude spectra of sqrt(I) * exp(i phi(depth)) for the rendered and the target
    image, phi = (2 pi / wavelength) * |depth - focal_depth|  (`TGD:392-425`)."""

    def __init__(self, wavelength: float = 0.05, focal_depth: float = 0.5):
        super().__init__()
        self.wavelength = wavelength
        self.focal_depth = focal_depth

    def forward(self, rendered: torch.Tensor, target: torch.Tensor, depth: torch.Tensor,
                wavelength: Optional[torch.Tensor] = None) -> torch.Tensor:
        lam = self.wavelength if wavelength is None else wavelength
        if depth.dim() == 4:  # (B,1,H,W) -> (B,H,W)
            depth = depth.squeeze(1)
        phi = ((2.0 * math.pi / lam) * (depth - self.focal_depth).abs()).unsqueeze(1)  # shared by the channels

        def spectrum_magnitude(img):
            amp = img.clamp(min=1e-8).sqrt()
            return torch.fft.fft2(torch.polar(amp, phi.expand_as(amp))).abs()

        return (spectrum_magnitude(rendered) - spectrum_magnitude(target)).square().mean()


class FrequencyDomainLoss(nn.Module):
    """MSE of the magnitude spectra below the radial cutoff plus `high_weight` times the MSE above it
    (`TGD:484-522`).  The two masks are complementary indicators, so both terms come from one pass:
    mean(w * (|F_r| - |F_t|)^2) with w = 1 below the cutoff and `high_weight` above."""

    def __init__(self, cutoff: float = 0.1, high_weight: float = 2.0):
        super().__init__()
        self.cutoff = cutoff
        self.high_weight = high_weight
        self._weights: Dict[Tuple[int, int, str], torch.Tensor] = {}

    def _weight(self, H: int, W: int, device: torch.device) -> torch.Tensor:
        key = (H, W, str(device))
        w = self._weights.get(key)
        if w is None:
            fy = torch.fft.fftfreq(H, device=device).unsqueeze(1)
            fx = torch.fft.fftfreq(W, device=device).unsqueeze(0)
            low = (fx * fx + fy * fy).sqrt() < self.cutoff
            w = torch.where(low, torch.ones((), device=device), torch.full((), float(self.high_weight), device=device))
            self._weights[key] = w
        return w

    def forward(self, rendered: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        H, W = rendered.shape[-2:]
        diff = torch.fft.fft2(rendered).abs() - torch.fft.fft2(target).abs()
        return (self._weight(H, W, rendered.device) * diff.square()).mean()


def wave_equation_loss(wave_field: torch.Tensor, wavelength: float, pixel_spacing: float = 1.0 / 256.0) -> torch.Tensor:
    """Mean squared Helmholtz residual lap(U) + k^2 U, k = 2 pi / wavelength, with the periodic 5-point
    Laplacian on a grid of spacing `pixel_spacing` (`TGD:781-835`).  (B,H,W) or (B,C,H,W)."""
    u = wave_field.unsqueeze(1) if wave_field.dim() == 3 else wave_field
    k2 = (2.0 * math.pi / wavelength) ** 2
    ring = u.roll(1, -1) + u.roll(-1, -1) + u.roll(1, -2) + u.roll(-1, -2)
    residual = (ring - 4.0 * u) / (pixel_spacing ** 2) + k2 * u
    return residual.square().mean()
