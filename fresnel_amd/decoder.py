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


def rotate_positions_for_pose(positions: torch.Tensor, elevation: torch.Tensor, azimuth: torch.Tensor) -> torch.Tensor:
    """View-aware rotation of the Gaussian grid (same maths as the reference's rotate_positions_for_pose,
    scripts/models/gaussian_decoder_models.py:51-104): azimuth about Y, then elevation about X, per image.
    positions (B, ..., 3); elevation / azimuth (B,) radians."""
    shape = (positions.shape[0],) + (1,) * (positions.dim() - 2)
    ca, sa = torch.cos(azimuth).view(shape), torch.sin(azimuth).view(shape)
    ce, se = torch.cos(elevation).view(shape), torch.sin(elevation).view(shape)
    x, y, z = positions[..., 0], positions[..., 1], positions[..., 2]
    xr = x * ca + z * sa
    zr = -x * sa + z * ca
    return torch.stack([xr, y * ce - zr * se, y * se + zr * ce], dim=-1)


class DepthEdgeDetector(nn.Module):
    """Edge strength in [0,1] of a (B,1,H,W) depth grid: Sobel gradients concatenated to the depth, three 3x3
    convolutions, sigmoid -- the layout of the reference's FresnelEdgeDetector (scripts/utils/fresnel_zones.py:
    1084-1160; its weights are learned, so only the structure is mirrored)."""

    def __init__(self, hidden_channels: int = 16):
        super().__init__()
        self.conv1 = nn.Conv2d(3, hidden_channels, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_channels, hidden_channels, 3, padding=1)
        self.conv3 = nn.Conv2d(hidden_channels, 1, 3, padding=1)
        self.register_buffer("sobel_x", torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]]).view(1, 1, 3, 3))
        self.register_buffer("sobel_y", torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]]).view(1, 1, 3, 3))

    def forward(self, depth: torch.Tensor) -> torch.Tensor:
        gx, gy = F.conv2d(depth, self.sobel_x, padding=1), F.conv2d(depth, self.sobel_y, padding=1)
        x = F.relu(self.conv1(torch.cat([depth, gx, gy], 1)))
        return torch.sigmoid(self.conv3(F.relu(self.conv2(x))))


class PatchGaussianDecoder(nn.Module):
    def __init__(self, feature_dim: int = 384, gaussians_per_patch: int = 4,
                 hidden_dims=(512, 512, 256, 128), grid: int = 37, use_fresnel_zones: bool = False,
                 num_fresnel_zones: int = 8, use_phase_output: bool = False, use_edge_aware: bool = False,
                 edge_scale_factor: float = 0.5, edge_opacity_boost: float = 0.2):
        super().__init__()
        # --use_edge_aware (TGD:1455-1461; decoder side gaussian_decoder_models.py:882-894): smaller, more opaque
        # Gaussians where the depth grid has edges
        self.use_edge_aware = use_edge_aware
        self.edge_scale_factor, self.edge_opacity_boost = edge_scale_factor, edge_opacity_boost
        self.edge_detector = DepthEdgeDetector() if use_edge_aware else None
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
        # identity-quaternion bias; a buffer so that the forward makes no host-to-device copy (not part of checkpoints)
        self.register_buffer("quat_bias", torch.tensor([1.0, 0.0, 0.0, 0.0]), persistent=False)

    def forward(self, features: torch.Tensor, depth: torch.Tensor, num_gaussians=None, elevation=None, azimuth=None, **_):
        """features (B,grid,grid,C) ; depth (B,1,h,w) in [0,1].
        num_gaussians: progressive growing (TGD:271-293, gaussian_decoder_models.py:770-790) -- the full capacity is
        predicted and only the first min(num_gaussians, K) Gaussians of every patch are used.
        elevation / azimuth (B,) radians: the grid is rotated to face the camera (rotate_positions_for_pose)."""
        Bn = features.shape[0]
        G, K = self.grid, self.gaussians_per_patch
        d = F.adaptive_avg_pool2d(depth, (G, G)).reshape(Bn, G * G, 1)
        x = torch.cat([features.reshape(Bn, G * G, -1), d], -1)
        o = self.mlp(x).reshape(Bn, G * G, K, self.per_gaussian)
        if num_gaussians is not None and num_gaussians < K:
            K = max(int(num_gaussians), 1)
            o = o[:, :, :K, :]
        cell = 2.0 / (G - 1)
        xy = self.grid_xy.view(1, G * G, 1, 2) + torch.tanh(o[..., 0:2]) * cell
        z = -2.0 - 2.0 * (d.unsqueeze(2) + 0.25 * torch.tanh(o[..., 2:3])).clamp(0, 1)
        if self.use_fresnel_zones:  # snap depth to zone centres (fresnel_zones.py:118-139), straight-through
            zq = -2.0 - 2.0 * ((torch.floor((-(z + 2.0) / 2.0).clamp(0, 0.999999) * self.num_fresnel_zones) + 0.5)
                               / self.num_fresnel_zones)
            z = z + (zq - z).detach()
        positions = torch.cat([xy, z.expand(-1, -1, K, -1)], -1)
        if elevation is not None and azimuth is not None:
            positions = rotate_positions_for_pose(positions, elevation, azimuth)
        scales = (0.13 + 0.03 * torch.sigmoid(o[..., 3:6])).clamp(1e-6, 2.0)
        opacities = torch.sigmoid(o[..., 13])
        if self.use_edge_aware:
            edge = self.edge_detector(d.reshape(Bn, 1, G, G)).reshape(Bn, G * G, 1)          # (B, G*G, 1) in [0,1]
            scales = scales * (1.0 - self.edge_scale_factor * edge.unsqueeze(-1))
            opacities = torch.clamp(opacities + self.edge_opacity_boost * edge, 0, 1)
        out = {
            "positions": positions.reshape(Bn, G * G * K, 3),
            "scales": scales.reshape(Bn, G * G * K, 3),
            "rotations": F.normalize(o[..., 6:10] + self.quat_bias, dim=-1
                                     ).reshape(Bn, G * G * K, 4),
            "colors": torch.sigmoid(o[..., 10:13]).reshape(Bn, G * G * K, 3),
            "opacities": opacities.reshape(Bn, G * G * K),
        }
        if self.use_phase_output:
            out["phases"] = torch.sigmoid(o[..., 15]).reshape(Bn, G * G * K)
        return out
