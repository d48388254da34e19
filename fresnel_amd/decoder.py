"""Patch decoder that PRODUCES renderer inputs (not a kernel target).

Own definition with the interface, shapes and ranges of the reference's DirectPatchDecoder
(scripts/models/gaussian_decoder_models.py:622-948): a 37x37 DINOv2 patch grid, K Gaussians per
patch -> dict{positions (B,N,3), scales (B,N,3) in [1e-6,2], rotations (B,N,4) unit wxyz,
colors/opacities in [0,1] [, phases (B,N) in [0,1]]}, N = 37*37*K (K=4 -> 5476).  It is the
module whose gradients the data-parallel step all-reduces (~0.63 M parameters at K=4).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PatchGaussianDecoder(nn.Module):
    def __init__(self, feature_dim: int = 384, gaussians_per_patch: int = 4,
                 hidden_dims=(512, 512, 256, 128), grid: int = 37, use_fresnel_zones: bool = False,
                 num_fresnel_zones: int = 8, use_phase_output: bool = False):
        super().__init__()
        self.grid = grid
        self.gaussians_per_patch = gaussians_per_patch
        self.use_fresnel_zones = use_fresnel_zones
        self.num_fresnel_zones = num_fresnel_zones
        self.use_phase_output = use_phase_output
        self.per_gaussian = 15 + (1 if use_phase_output else 0)  # pos3 scale3 quat4 color3 opa1 (+phase)
        dims = [feature_dim + 1] + list(hidden_dims)
        layers = []
        for a, b in zip(dims[:-1], dims[1:]):
            layers += [nn.Linear(a, b), nn.GELU()]
        layers.append(nn.Linear(dims[-1], gaussians_per_patch * self.per_gaussian))
        self.mlp = nn.Sequential(*layers)
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, grid), torch.linspace(-1, 1, grid), indexing="ij")
        self.register_buffer("grid_xy", torch.stack([xs, -ys], -1).reshape(grid * grid, 2))

    def forward(self, features: torch.Tensor, depth: torch.Tensor, num_gaussians=None, **_):
        """features (B,grid,grid,C) ; depth (B,1,h,w) in [0,1]."""
        Bn = features.shape[0]
        G, K = self.grid, self.gaussians_per_patch
        d = F.adaptive_avg_pool2d(depth, (G, G)).reshape(Bn, G * G, 1)
        x = torch.cat([features.reshape(Bn, G * G, -1), d], -1)
        o = self.mlp(x).reshape(Bn, G * G, K, self.per_gaussian)
        cell = 2.0 / (G - 1)
        xy = self.grid_xy.view(1, G * G, 1, 2) + torch.tanh(o[..., 0:2]) * cell
        z = -2.0 - 2.0 * (d.unsqueeze(2) + 0.25 * torch.tanh(o[..., 2:3])).clamp(0, 1)
        if self.use_fresnel_zones:  # snap depth to zone centres (fresnel_zones.py:118-139), straight-through
            zq = -2.0 - 2.0 * ((torch.floor((-(z + 2.0) / 2.0).clamp(0, 0.999999) * self.num_fresnel_zones) + 0.5)
                               / self.num_fresnel_zones)
            z = z + (zq - z).detach()
        out = {
            "positions": torch.cat([xy, z], -1).reshape(Bn, G * G * K, 3),
            "scales": (0.13 + 0.03 * torch.sigmoid(o[..., 3:6])).clamp(1e-6, 2.0).reshape(Bn, G * G * K, 3),
            "rotations": F.normalize(o[..., 6:10] + torch.tensor([1.0, 0, 0, 0], device=o.device), dim=-1
                                     ).reshape(Bn, G * G * K, 4),
            "colors": torch.sigmoid(o[..., 10:13]).reshape(Bn, G * G * K, 3),
            "opacities": torch.sigmoid(o[..., 13]).reshape(Bn, G * G * K),
        }
        if self.use_phase_output:
            out["phases"] = torch.sigmoid(o[..., 15]).reshape(Bn, G * G * K)
        return out
