"""Decoder -> rasterizer hand-off: what the reference's train_epoch does between the decoder's output
dict and the renderer call, restated for the batched HIP renderer.  Reference = scripts/training/
train_gaussian_decoder.py ("TGD"):

    HFTSConfig                      TGD:239-303   training-speed schedule: render resolution, progressive
                                                  Gaussians-per-patch, stochastic K
    importance_subsample            TGD:1154-1187 K Gaussians drawn with p ~ mean opacity, without replacement, the
                                                  same indices for every image of the batch
    sample_training_pose            TGD:1078-1098 one pose per batch: frontal with probability frontal_prob, else
                                                  uniform in the elevation / azimuth ranges
    camera_for_batch                TGD:1196-1207 frontal camera, or create_camera_from_pose for a novel view

The gather itself runs in libfgs_hip.so (fgs_gather_forward / fgs_gather_backward: one launch for all tensors); the
draw is torch.multinomial on the device, as in the reference.
"""
import ctypes
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _binding as B


@dataclass
class HFTSConfig:
    """Same fields, defaults and methods as the reference's HFTSConfig (TGD:239-303)."""
    train_resolution: Optional[int] = None   # render / loss resolution (None = image_size)
    progressive_schedule: bool = False       # progressive Gaussian growing
    stochastic_k: Optional[int] = None       # sample K Gaussians (None = all)
    fast_mode: bool = False                  # all of the above with the preset values

    def get_effective_train_resolution(self, image_size: int) -> int:
        if self.fast_mode:
            return 64
        return self.train_resolution if self.train_resolution is not None else image_size

    def get_gaussians_per_patch(self, epoch: int, total_epochs: int, base_gpp: int = 4) -> int:
        """1 / 2 / max(4, base) / base Gaussians per patch over the four quarters of training."""
        if not self.progressive_schedule and not self.fast_mode:
            return base_gpp
        progress = epoch / max(total_epochs, 1)
        if progress < 0.25:
            return 1
        if progress < 0.50:
            return 2
        if progress < 0.75:
            return max(4, base_gpp)
        return base_gpp

    def get_stochastic_k(self, total_gaussians: int) -> int:
        """Number of Gaussians to render: K (256 in fast mode), or all of them."""
        if self.fast_mode and self.stochastic_k is None:
            return min(256, total_gaussians)
        if self.stochastic_k is not None:
            return min(self.stochastic_k, total_gaussians)
        return total_gaussians

    @property
    def enabled(self) -> bool:
        return (self.fast_mode or self.train_resolution is not None or self.progressive_schedule
                or self.stochastic_k is not None)


_KEYS = ("positions", "scales", "rotations", "colors", "opacities")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _GatherGaussians(torch.autograd.Function):
    """(B,N,.) x indices (K,) -> (B,K,.) for all Gaussian tensors in one launch (fgs_gather_forward/backward)."""

    @staticmethod
    def forward(ctx, indices, pos, scale, quat, color, opacity, phase):
        if not pos.is_cuda:
            raise B.FgsError("importance_subsample (HIP gather) needs CUDA/ROCm tensors; there is no CPU fallback")
        lib = B.load()
        Bn, N, K = pos.shape[0], pos.shape[1], indices.shape[0]
        ins = [t.detach().contiguous().float() for t in (pos, scale, quat, color, opacity)]
        ph = phase.detach().contiguous().float() if phase is not None else None
        if ph is not None and not (ph.dim() == 2 or (ph.dim() == 3 and ph.shape[2] in (1, 3))):
            raise ValueError(f"phases must be (B,N) or (B,N,3), got {tuple(ph.shape)}")
        pc = 0 if ph is None else (int(ph.shape[2]) if ph.dim() == 3 else 1)  # floats per Gaussian the kernel moves
        idx = indices.detach().to(torch.int64).contiguous()
        dev = pos.device
        with torch.cuda.device(dev):
            outs = [torch.empty(Bn, K, *t.shape[2:], dtype=torch.float32, device=dev) for t in ins]
            o_ph = torch.empty(Bn, K, *ph.shape[2:], dtype=torch.float32, device=dev) if ph is not None else None
            B.check(lib.fgs_gather_forward(Bn, N, K, pc, _ptr(idx), *[_ptr(t) for t in ins], _ptr(ph),
                                           *[_ptr(t) for t in outs], _ptr(o_ph),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "fgs_gather_forward")
        ctx.save_for_backward(idx)
        ctx.shape = (Bn, N, K, pc)
        ctx.trail = [t.shape[2:] for t in ins] + [ph.shape[2:] if ph is not None else None]
        return tuple(outs) + (o_ph,)

    @staticmethod
    def backward(ctx, *grads):
        lib = B.load()
        (idx,) = ctx.saved_tensors
        Bn, N, K, pc = ctx.shape
        dev = idx.device
        with torch.cuda.device(dev):
            gs = []
            for g, trail in zip(grads[:5], ctx.trail[:5]):
                gs.append((g if g is not None else torch.zeros(Bn, K, *trail, device=dev)).contiguous().float())
            g_ph = None
            if pc:
                g_ph = (grads[5] if grads[5] is not None else torch.zeros(Bn, K, *ctx.trail[5], device=dev)).contiguous().float()
            outs = [torch.empty(Bn, N, *trail, dtype=torch.float32, device=dev) for trail in ctx.trail[:5]]
            o_ph = torch.empty(Bn, N, *ctx.trail[5], dtype=torch.float32, device=dev) if pc else None
            B.check(lib.fgs_gather_backward(Bn, N, K, pc, _ptr(idx), *[_ptr(t) for t in gs], _ptr(g_ph),
                                            *[_ptr(t) for t in outs], _ptr(o_ph),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "fgs_gather_backward")
        return (None,) + tuple(outs) + (o_ph,)


def importance_weights(opacities: torch.Tensor) -> torch.Tensor:
    """p(i) ~ mean opacity of Gaussian i over the batch, + 1e-6, normalised (TGD:1166-1170)."""
    w = opacities.detach().mean(dim=0) + 1e-6
    return w / w.sum()


def importance_subsample(output: Dict[str, torch.Tensor], k: Optional[int],
                         generator: Optional[torch.Generator] = None,
                         indices: Optional[torch.Tensor] = None) -> Tuple[Dict[str, torch.Tensor], Optional[torch.Tensor]]:
    """Stochastic Gaussian rendering (TGD:1154-1187): keep K of the N Gaussians of every image, drawn without
    replacement with probability proportional to the batch-mean opacity; the same K indices for the whole batch.
    Returns (subsampled dict, indices) -- the dict unchanged and None when k is None or k >= N.
    `indices` overrides the draw (tests)."""
    N = output["positions"].shape[1]
    if k is None or k >= N:
        return output, None
    if indices is None:
        with torch.no_grad():
            indices = torch.multinomial(importance_weights(output["opacities"]), k, replacement=False,
                                        generator=generator)
    phases = output.get("phases")
    outs = _GatherGaussians.apply(indices, *[output[key] for key in _KEYS], phases)
    sampled = dict(zip(_KEYS, outs[:5]))
    if phases is not None:
        sampled["phases"] = outs[5]
    return sampled, indices


def sample_training_pose(multi_pose_augmentation: bool, use_pose_encoding: bool, frontal_prob: float,
                         pose_range_elevation=(-30.0, 45.0), pose_range_azimuth=(0.0, 360.0),
                         rng: Optional[np.random.RandomState] = None) -> Tuple[Optional[float], Optional[float], bool]:
    """One (elevation, azimuth) in radians for the whole batch (TGD:1078-1098): (None, None, True) when pose
    augmentation is off; frontal (0, 0) with probability frontal_prob; otherwise uniform in the given degree
    ranges.  Draw order as in the reference: frontal decision, elevation, azimuth."""
    if not (multi_pose_augmentation and use_pose_encoding):
        return None, None, True
    rng = rng if rng is not None else np.random
    if rng.random_sample() < frontal_prob:
        return 0.0, 0.0, True
    el = rng.uniform(math.radians(pose_range_elevation[0]), math.radians(pose_range_elevation[1]))
    az = rng.uniform(math.radians(pose_range_azimuth[0]), math.radians(pose_range_azimuth[1]))
    return float(el), float(az), False


def camera_for_batch(frontal_camera, elevation: Optional[float], azimuth: Optional[float], render_size: int,
                     multi_pose_augmentation: bool):
    """The camera the batch is rendered with (TGD:1196-1207): the frontal one, or an orbit camera at the sampled
    pose (focal_length_mult 0.8 like the frontal set-up)."""
    if multi_pose_augmentation and elevation is not None and azimuth is not None:
        from .renderer import create_camera_from_pose
        return create_camera_from_pose(elevation, azimuth, render_size, focal_length_mult=0.8)
    return frontal_camera
