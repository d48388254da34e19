#!/usr/bin/env python3
"""Training harness around the HIP rasterizer: this repo's counterpart of the reference's
scripts/training/train_gaussian_decoder.py ("TGD") for the path BASELINE.json names.

Kept from TGD (same flag names / defaults / behaviour):
  flags             --experiment --data_dir --output_dir --batch_size --epochs --lr --image_size
                    --gaussians_per_patch --max_images --use_fresnel_zones --num_fresnel_zones
                    --use_phase_blending --phase_amplitude --resume            TGD:1401-1545
                    --use_edge_aware --edge_scale_factor --stochastic_k --progressive_schedule --train_resolution
                    --fast_mode --multi_pose_augmentation --pose_range_elevation/azimuth --frontal_prob
                    --use_pose_encoding                                         TGD:1455-1461, 1524-1545
  data              ImageDataset over --data_dir with the `features/` caches (fresnel_amd/data.py, TGD:525-675);
                    synthetic stand-in data only when --data_dir holds no images
  hand-off          progressive Gaussians-per-patch, importance subsampling of K Gaussians, one orbit camera per
                    batch (fresnel_amd/handoff.py, TGD:1069-1207)
  renderer choice   TileBasedRenderer(res, res, use_phase_blending, phase_amplitude)   TGD:1898-1907
  camera            fx = fy = 0.8*res, cx = cy = res/2, view = I                       TGD:1910-1917
  step              decoder -> render -> stack -> L1 + normalised-depth L1 (SSIM / LPIPS only when
                    those packages exist, as TGD:53-65) -> NaN/Inf skip -> backward ->
                    clip_grad_norm_(1.0) -> AdamW(lr, weight_decay=1e-5) ; CosineAnnealingLR
                                                                    TGD:890, 922-930, 1255-1266, 1970-1971
  data order        a fresh permutation of the images every epoch (DataLoader(shuffle=True), TGD:1760-1767), drawn
                    from seed + epoch so that every rank draws the same one; the last, partial batch is kept when it
                    divides over the ranks; batches are decoded ahead of the step by worker processes / a prefetch thread
                    into pinned memory
  checkpoints       {epoch, model_state_dict, optimizer_state_dict, losses, config}    TGD:1304-1310
  metrics           print every log_interval batches (TGD:1275-1281); training_history_exp{E}.json (TGD:1317-1323) with
                    the per-epoch losses plus step ms, composited Gaussian-pixels/s and the rasterizer's stage ms
Changed on purpose:
  * the per-image Python loop TGD:1209-1223 becomes ONE batched renderer call;
  * NO host synchronisation inside a step: loss terms stay on the device and are fetched once per log_interval; the
    NaN/Inf batch skip (TGD:1255-1258) is decided on the device -- the flag rides in the gradient bucket's all-reduce
    and the fused AdamW step takes it as `found_inf` (weights, moments and step count untouched, exactly a skipped
    batch).  The reference syncs ~5 times per Gaussian (.item(), DR:584-597) plus once per loss term;
  * image-wise data parallelism: one process per GPU, the batch is sharded by image and the
    decoder gradients are all-reduced once per step over RCCL (fresnel_amd/dist.py);
  * no dataset preprocessing tooling (DINOv2 / depth extraction): caches are read when present; when --data_dir has
    no images the harness fabricates images/features like TGD:1748-1758 / TGD:613-630 do when files are missing.

    python -m fresnel_amd.train --experiment 2 --epochs 1                      # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 -m fresnel_amd.train    # 8 GPUs, DP
"""
import argparse
import inspect
import json
import math
import os
import queue
import threading
import time
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F
from torch.optim import AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR

import numpy as np

from .decoder import PatchGaussianDecoder
from .dist import DPContext
from .handoff import HFTSConfig, camera_for_batch, importance_subsample, sample_training_pose

try:  # optional perceptual losses, exactly like TGD:53-65
    from pytorch_msssim import ssim as ssim_fn
    SSIM_AVAILABLE = True
except ImportError:
    SSIM_AVAILABLE = False


@dataclass
class TrainingConfig:  # subset of TGD:97-162 that this path uses; same names and defaults
    experiment: int = 2
    data_dir: str = "images"
    output_dir: str = "checkpoints"
    batch_size: int = 4
    epochs: int = 100
    lr: float = 1e-4
    weight_decay: float = 1e-5
    image_size: int = 256
    feature_size: int = 37
    feature_dim: int = 384
    rgb_weight: float = 1.0
    depth_weight: float = 0.1
    ssim_weight: float = 0.5
    gaussians_per_patch: int = 4
    max_images: Optional[int] = None
    use_fresnel_zones: bool = False
    num_fresnel_zones: int = 8
    use_phase_blending: bool = False
    phase_amplitude: float = 0.25
    use_edge_aware: bool = False      # smaller / more opaque Gaussians at depth edges (TGD:147-151)
    edge_scale_factor: float = 0.5
    edge_opacity_boost: float = 0.2
    multi_pose_augmentation: bool = False  # one random orbit pose per batch (TGD:155-159)
    pose_range_elevation: tuple = (-30.0, 45.0)
    pose_range_azimuth: tuple = (0.0, 360.0)
    frontal_prob: float = 0.3
    use_pose_encoding: bool = False
    use_wave_rendering: bool = False  # WaveFieldRenderer (TGD:178, 1891-1897); --use_qsr implies it
    # spectral / stencil losses after the renderer (TGD:191, 225-229; fresnel_amd/losses.py)
    wave_equation_weight: float = 0.0
    wavelength: float = 0.05
    use_phase_retrieval_loss: bool = False
    phase_retrieval_weight: float = 0.1
    use_frequency_loss: bool = False
    frequency_loss_weight: float = 0.1
    device: str = "cuda" if torch.cuda.is_available() else "cpu"
    log_interval: int = 10
    save_interval: int = 10
    seed: int = 0
    steps_per_epoch: int = 8  # synthetic-data mode only
    num_workers: int = 4      # host threads decoding the NEXT batches (DataLoader(num_workers=4), TGD:1760-1767)
    hip_graph: bool = False   # replay the whole step (decoder, rasterizer, losses, backward, all-reduce, clip, AdamW) from
                              # ONE captured HIP graph: the library never allocates or synchronises and the step has no
                              # host decision left, so it is capturable; pays when the step is host-launch-bound


class SyntheticDataset:
    """Stand-in for TGD's ImageDataset when no caches exist: random images, zero-mean random
    features, smooth random depth (TGD:613-630 returns zero features/depth for missing caches;
    TGD:1748-1758 fabricates random PNGs).  Deterministic in the GLOBAL sample index so every DP
    configuration sees the same data."""

    def __init__(self, n_items, cfg: TrainingConfig):
        self.n, self.cfg = n_items, cfg

    def get(self, idx):
        g = torch.Generator().manual_seed(self.cfg.seed * 1_000_003 + idx)
        S, Fs = self.cfg.image_size, self.cfg.feature_size
        low = torch.rand(3, 8, 8, generator=g)
        image = F.interpolate(low[None], size=(S, S), mode="bilinear", align_corners=False)[0]
        feats = torch.randn(Fs, Fs, self.cfg.feature_dim, generator=g) * 0.5
        dlow = torch.rand(1, 4, 4, generator=g)
        depth = F.interpolate(dlow[None], size=(S, S), mode="bilinear", align_corners=False)[0]
        return image, feats, depth

    def __len__(self):
        return self.n

    def host_item(self, idx):
        return self.get(idx)

    def batch(self, indices, device):
        items = [self.get(i) for i in indices]
        return tuple(torch.stack(t).to(device) for t in zip(*items))


def epoch_batches(n_items: int, batch_size: int, world: int, seed: int, epoch: int):
    """Global batches of one epoch as index lists: a fresh permutation per epoch (DataLoader(shuffle=True),
    TGD:1760-1767) from a generator seeded with seed + epoch -- every rank draws the SAME permutation and takes its
    shard of each batch.  The last, partial batch is kept, as the reference keeps it, when it divides over the ranks
    (equal shards: the mean of shard means stays the global mean); otherwise it is dropped."""
    perm = torch.randperm(n_items, generator=torch.Generator().manual_seed(seed + epoch)).tolist()
    out = [perm[i:i + batch_size] for i in range(0, n_items - batch_size + 1, batch_size)]
    tail = perm[len(out) * batch_size:]
    if tail and len(tail) % world == 0:
        out.append(tail)
    return out


class BatchPrefetcher:
    """Iterates device batches for a list of index lists, decoding AHEAD of the training step: `num_workers` host
    threads load items (PIL decode / LANCZOS resize / cache reads release the GIL), a producer thread stacks them into
    pinned memory, at most `depth` batches wait in the queue, and the copy to the GPU is asynchronous.  The reference
    gets the same overlap from DataLoader(num_workers=4, pin_memory=True) worker processes (TGD:1760-1767)."""

    def __init__(self, data, index_lists, device, num_workers=4, depth=2):
        self.data, self.lists, self.device = data, index_lists, torch.device(device)
        self.q = queue.Queue(maxsize=max(1, depth))
        self.pin = self.device.type == "cuda"
        self.workers = max(1, int(num_workers))
        self.stop = threading.Event()
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    def _put(self, item):
        """Queue `item` unless the consumer has gone away (close()): never blocks for good."""
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def _produce(self):
        from concurrent.futures import ThreadPoolExecutor
        try:
            with ThreadPoolExecutor(self.workers) as pool:
                for idx in self.lists:
                    if self.stop.is_set():
                        return
                    items = list(pool.map(self.data.host_item, idx))
                    batch = tuple(torch.stack(t) for t in zip(*items))
                    if self.pin:
                        batch = tuple(t.pin_memory() for t in batch)
                    if not self._put(batch):
                        return
            self._put(None)
        except BaseException as e:  # surfaces in the consumer, not in a dead thread
            self._put(e)

    def close(self):
        """Stop the producer and drop what it queued (pinned batches).  Called when the consumer leaves the loop for
        any reason -- exhaustion, `break`, an exception in the training step -- so no thread is left blocked in `put`."""
        self.stop.set()
        while True:
            try:
                self.q.get_nowait()
            except queue.Empty:
                break
        self.thread.join(timeout=5.0)

    def __iter__(self):
        try:
            while True:
                b = self.q.get()
                if b is None:
                    return
                if isinstance(b, BaseException):
                    raise b
                yield tuple(t.to(self.device, non_blocking=True) for t in b)
        finally:
            self.close()


def _global_mean_std(x, dp):
    """mean and unbiased std (torch.std default, as TGD:922-925 uses) of x over the GLOBAL batch, TWO-PASS like
    torch.std: the sum is all-reduced for the mean, then the sum of squared deviations (the one-pass
    sum x^2 - n mean^2 loses digits to cancellation for ~2 M depth values whose mean is far from zero).  Both
    all-reduces are differentiable, so an N-rank step normalises the depth loss exactly like the 1-rank step."""
    dist_on = dp is not None and dp.enabled
    n = x.numel() * (dp.world if dist_on else 1)
    s1 = x.sum()
    if dist_on:
        s1 = dp.global_sum(s1.reshape(1))[0]
    mean = s1 / n
    s2 = ((x - mean) ** 2).sum()
    if dist_on:
        s2 = dp.global_sum(s2.reshape(1))[0]
    return mean, torch.sqrt(s2 / max(n - 1, 1))


def compute_losses(rendered, target, rendered_depth, target_depth, cfg: TrainingConfig, dp=None):
    """L1 + (1-SSIM if available) + normalised depth L1 (TGD:890, 906-930).  Returns (total, terms): `terms` holds the
    individual losses as DETACHED 0-d DEVICE tensors -- nothing here synchronises with the host (the reference's
    `.item()` per term, TGD:891-1001, would cost a device round trip each)."""
    d: Dict[str, torch.Tensor] = {}
    rgb = F.l1_loss(rendered, target)
    d["rgb"] = rgb.detach()
    total = cfg.rgb_weight * rgb
    if SSIM_AVAILABLE and cfg.ssim_weight > 0:
        s = 1.0 - ssim_fn(torch.clamp(rendered, 0, 1), target, data_range=1.0, size_average=True)
        d["ssim"] = s.detach()
        total = total + cfg.ssim_weight * s
    if rendered_depth is not None and target_depth is not None:
        rd_mean, rd_std = _global_mean_std(rendered_depth, dp)
        td_mean, td_std = _global_mean_std(target_depth, dp)
        dl = F.l1_loss((rendered_depth - rd_mean) / torch.clamp(rd_std, min=1e-4),
                       (target_depth - td_mean) / torch.clamp(td_std, min=1e-4))
        d["depth"] = dl.detach()
        total = total + cfg.depth_weight * dl
    if cfg.wave_equation_weight > 0:  # TGD:957-964
        from .losses import wave_equation_loss
        we = wave_equation_loss(rendered, cfg.wavelength, pixel_spacing=1.0 / cfg.image_size)
        d["wave_eq"] = we.detach()
        total = total + cfg.wave_equation_weight * we
    if cfg.use_phase_retrieval_loss and target_depth is not None:  # TGD:972-983
        pr = _loss_module("phase", cfg)(rendered, target, target_depth)
        d["phase_retrieval"] = pr.detach()
        total = total + cfg.phase_retrieval_weight * pr
    if cfg.use_frequency_loss:  # TGD:990-996
        fq = _loss_module("freq", cfg)(rendered, target)
        d["frequency"] = fq.detach()
        total = total + cfg.frequency_loss_weight * fq
    d["total"] = total.detach()
    return total, d


_LOSS_MODULES: Dict[str, torch.nn.Module] = {}


def _loss_module(kind: str, cfg: TrainingConfig):
    """One PhaseRetrievalLoss / FrequencyDomainLoss per process (TGD:1683-1690 builds them once)."""
    if kind not in _LOSS_MODULES:
        from .losses import FrequencyDomainLoss, PhaseRetrievalLoss
        _LOSS_MODULES[kind] = PhaseRetrievalLoss(wavelength=cfg.wavelength) if kind == "phase" else FrequencyDomainLoss()
    return _LOSS_MODULES[kind]


def default_renderer_factory(cfg: TrainingConfig, device, res: Optional[int] = None):
    """The product renderer: HIP TileBasedRenderer + the reference camera (TGD:1898-1917), at the effective training
    resolution (HFTS --train_resolution / --fast_mode, TGD:1643, 1880-1887)."""
    from .renderer import Camera, TileBasedRenderer
    res = res or cfg.image_size
    if cfg.use_wave_rendering:  # TGD:1891-1897
        from .renderer import WaveFieldRenderer
        renderer = WaveFieldRenderer(res, res).to(device)
    else:                       # TGD:1898-1906
        renderer = TileBasedRenderer(res, res, use_phase_blending=cfg.use_phase_blending,
                                     phase_amplitude=cfg.phase_amplitude).to(device)
        if cfg.use_fresnel_zones:
            # the decoder snaps depths to `num_fresnel_zones` values (GDM:834-841): the zone-key depth sort then needs one
            # radix pass instead of four (FgsDims.sort_mode, fgs_sort.hip); a work-split choice, the order is the same
            renderer.tuning = dict(renderer.tuning or {})
            renderer.tuning.setdefault("sort_mode", 1)  # (merged: an existing tuning, or an explicit sort_mode=0, is kept -- ADVICE r4)
    camera = Camera(fx=res * 0.8, fy=res * 0.8, cx=res / 2, cy=res / 2, width=res, height=res)
    return renderer, camera


class StepResult:
    """What one train_step leaves behind, all on the device: `terms` (name -> 0-d tensor, this rank's shard; "total"
    is the mean over the ranks) and `skipped` (0-d float: 1.0 when the global batch was skipped for a NaN/Inf loss).
    `to_host()` is the only place that synchronises."""

    def __init__(self, terms, skipped):
        self.terms, self.skipped = terms, skipped

    def to_host(self):
        keys = sorted(self.terms)
        vals = torch.stack([self.terms[k].float() for k in keys] + [self.skipped.float()]).tolist()  # one transfer
        if vals[-1] > 0:
            return None
        return dict(zip(keys, vals[:-1]))


def make_optimizer(model, cfg: TrainingConfig):
    """AdamW(lr, weight_decay) as TGD:1970, FUSED: one multi-tensor kernel per step, and it takes the skip flag as a
    device tensor (`found_inf`, the mechanism GradScaler uses), so the NaN/Inf batch skip needs no host decision.
    With cfg.hip_graph the learning rate is a device tensor too (capturable): the cosine schedule then reaches a
    replayed graph by an in-place fill instead of a host scalar baked into the capture."""
    params = list(model.parameters())
    try:
        if cfg.hip_graph and params and params[0].is_cuda:
            return AdamW(params, lr=torch.tensor(cfg.lr, device=params[0].device), weight_decay=cfg.weight_decay,
                         fused=True, capturable=True)
        return AdamW(params, lr=cfg.lr, weight_decay=cfg.weight_decay, fused=True)
    except (RuntimeError, TypeError):  # a device without the fused kernel: plain AdamW, the skip costs one sync
        return AdamW(params, lr=cfg.lr, weight_decay=cfg.weight_decay)


class GraphedTrainStep:
    """train_step captured ONCE in a HIP graph (torch.cuda.CUDAGraph) and replayed per batch: the step's ~200 kernel
    launches (decoder MLP, the rasterizer's ~20 kernels each way, losses, clip, fused AdamW, the all-reduce) cost one
    graph launch on the host.  Possible because nothing in the step allocates behind torch's back, synchronises or takes
    a host decision (the NaN/Inf skip is `found_inf` on the device).  Inputs are copied into static buffers; the
    StepResult's tensors are static too (overwritten by the next replay -- read them before it).
    Eager fallback (`matches` is False): a batch of another shape (the epoch's tail batch)."""

    def __init__(self, model, renderer, camera, optimizer, cfg, dp, example_batch, epoch=0, train_res=None):
        self.static = tuple(torch.empty_like(t) for t in example_batch)
        for s_, t in zip(self.static, example_batch):
            s_.copy_(t)
        args = (model, renderer, camera, self.static, optimizer, cfg, dp)
        kw = dict(hfts=None, epoch=epoch, train_res=train_res, pose_rng=None, sample_gen=None)
        # warm-up on a side stream (plan caches, optimizer state, allocator), then put parameters and optimizer state
        # back: the capture must not cost the run three optimizer steps
        params = [p.detach().clone() for p in model.parameters()]
        had_state = len(optimizer.state) > 0
        saved_state = None
        if had_state:
            saved_state = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                           for p, st in optimizer.state.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                train_step(*args, **kw)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for p, q in zip(model.parameters(), params):
                p.copy_(q)
            for p, st in optimizer.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.copy_(saved_state[id(p)][k]) if had_state else v.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = train_step(*args, **kw)

    def matches(self, batch) -> bool:
        return all(s_.shape == t.shape and s_.dtype == t.dtype for s_, t in zip(self.static, batch))

    def __call__(self, batch):
        for s_, t in zip(self.static, batch):
            s_.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.result


def train_step(model, renderer, camera, batch, optimizer, cfg: TrainingConfig, dp: DPContext,
               hfts: Optional[HFTSConfig] = None, epoch: int = 0, train_res: Optional[int] = None,
               pose_rng: Optional[np.random.RandomState] = None, sample_gen: Optional[torch.Generator] = None):
    """One optimizer step on this rank's image shard, without a host synchronisation.  Returns a StepResult."""
    images, feats, depth = batch
    res = train_res or cfg.image_size
    # progressive Gaussian growing (TGD:1069-1076) and the batch's pose (TGD:1078-1098)
    num_gaussians = hfts.get_gaussians_per_patch(epoch, cfg.epochs, cfg.gaussians_per_patch) if hfts is not None else None
    el, az, _ = sample_training_pose(cfg.multi_pose_augmentation, cfg.use_pose_encoding, cfg.frontal_prob,
                                     cfg.pose_range_elevation, cfg.pose_range_azimuth, pose_rng)
    el_t = torch.full((feats.shape[0],), el, device=feats.device) if el is not None else None
    az_t = torch.full((feats.shape[0],), az, device=feats.device) if az is not None else None
    out = model(feats, depth, num_gaussians=num_gaussians, elevation=el_t, azimuth=az_t)
    if not (cfg.use_phase_blending or cfg.use_wave_rendering):
        out.pop("phases", None)
    # stochastic Gaussian rendering: K Gaussians by opacity importance, gathered on the device (TGD:1154-1187)
    if hfts is not None:
        out, _ = importance_subsample(out, hfts.get_stochastic_k(out["positions"].shape[1]), generator=sample_gen)
    phases = out.get("phases")
    if cfg.use_wave_rendering and phases is not None:
        phases = phases * (2.0 * math.pi)  # wave renderers take radians (DR:772), the decoder emits [0,1]
    render_camera = camera_for_batch(camera, el, az, res, cfg.multi_pose_augmentation)  # TGD:1196-1207
    # ONE batched call replaces the per-image loop of TGD:1209-1223
    rendered, rdepth = renderer(out["positions"], out["scales"], out["rotations"], out["colors"],
                                out["opacities"], render_camera, return_depth=True, phases=phases)
    target = F.interpolate(images, size=(res, res), mode="bilinear", align_corners=False)
    tdepth = F.interpolate(depth, size=(res, res), mode="bilinear", align_corners=False).squeeze(1)
    loss, terms = compute_losses(rendered, target, rdepth, tdepth, cfg, dp)
    # NaN/Inf batch skip (TGD:1255-1258), decided on the device and by ALL ranks together: the flag and the loss ride
    # behind the gradients in the one all-reduce of the step
    bad = (~torch.isfinite(loss.detach())).float()
    if dp.enabled:
        dp.adopt(list(model.parameters()))  # .grad = slices of the flat bucket, zeroed in one launch: nothing to pack later
    else:
        optimizer.zero_grad(set_to_none=True)
    loss.backward()
    extra = torch.stack([bad, torch.nan_to_num(loss.detach().float(), nan=0.0, posinf=0.0, neginf=0.0)])
    extra = dp.allreduce_gradients(list(model.parameters()), extra=extra)
    skipped = (extra[0] > 0).float()
    terms["total"] = extra[1] / dp.world
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)  # TGD:1264 (foreach, no host sync)
    if getattr(optimizer, "_step_supports_amp_scaling", False) and optimizer.defaults.get("fused"):
        optimizer.grad_scale, optimizer.found_inf = None, skipped.reshape(())  # skipped batch: the fused step is a no-op
        optimizer.step()
    elif not bool(skipped):  # non-fused optimizer: the skip is a host decision (one sync)
        optimizer.step()
    return StepResult(terms, skipped)


def save_checkpoint(model, optimizer, epoch, losses, cfg: TrainingConfig):
    os.makedirs(cfg.output_dir, exist_ok=True)
    ckpt = {"epoch": epoch, "model_state_dict": model.state_dict(),
            "optimizer_state_dict": optimizer.state_dict(), "losses": losses, "config": asdict(cfg)}
    path = Path(cfg.output_dir) / f"decoder_exp{cfg.experiment}_epoch{epoch}.pt"
    torch.save(ckpt, path)
    return path


def make_dataset(cfg: TrainingConfig, log=print):
    """ImageDataset over --data_dir (with its `features/` caches) when it holds images; otherwise the synthetic
    stand-in (the reference fabricates test images itself when the directory is empty, TGD:1748-1758)."""
    if cfg.data_dir and os.path.isdir(cfg.data_dir):
        from .data import ImageDataset
        ds = ImageDataset(cfg.data_dir, cfg.image_size, max_images=cfg.max_images, feature_dim=cfg.feature_dim)
        if len(ds) > 0:
            if ds[0]["features"].shape[-1] != cfg.feature_size:
                raise ValueError(f"feature caches are 37x37 patch grids; got feature_size={cfg.feature_size}")
            log(f"Found {len(ds)} images in {cfg.data_dir}")
            return ds, len(ds)
        log(f"No images in {cfg.data_dir}: synthetic data")
    n_items = cfg.max_images or cfg.batch_size * cfg.steps_per_epoch
    return SyntheticDataset(n_items, cfg), n_items


def _make_renderer(renderer_factory, cfg, device, train_res):
    """Factories take (cfg, device) or (cfg, device, res): chosen from the signature (a TypeError raised INSIDE a
    three-argument factory must not be mistaken for a two-argument one)."""
    try:
        params = inspect.signature(renderer_factory).parameters.values()
        takes_res = len(params) >= 3 or any(p.kind == p.VAR_POSITIONAL for p in params)
    except (TypeError, ValueError):
        takes_res = True
    return renderer_factory(cfg, device, train_res) if takes_res else renderer_factory(cfg, device)


def save_training_history(history: Dict[str, list], cfg: TrainingConfig):
    """training_history_exp{E}.json (TGD:1317-1323): one list per loss term, one entry per epoch, plus this repo's
    step ms, composited Gaussian-pixels/s (whole job) and the rasterizer's stage ms."""
    os.makedirs(cfg.output_dir, exist_ok=True)
    path = Path(cfg.output_dir) / f"training_history_exp{cfg.experiment}.json"
    with open(path, "w") as f:
        json.dump(history, f, indent=2)
    return path


def run_training(cfg: TrainingConfig, dp: Optional[DPContext] = None,
                 renderer_factory: Callable = default_renderer_factory, resume: Optional[str] = None,
                 log=print, hfts: Optional[HFTSConfig] = None):
    device = torch.device(cfg.device)
    dp = dp or DPContext(device=device if device.type == "cuda" else None)
    if cfg.batch_size % dp.world != 0:  # fail fast, before any rank can stall in a collective (see DPContext.shard)
        raise ValueError(f"--batch_size {cfg.batch_size} is not a multiple of the {dp.world} ranks")
    torch.manual_seed(cfg.seed)
    model = PatchGaussianDecoder(cfg.feature_dim, cfg.gaussians_per_patch, grid=cfg.feature_size,
                                 use_fresnel_zones=cfg.use_fresnel_zones,
                                 num_fresnel_zones=cfg.num_fresnel_zones,
                                 use_phase_output=cfg.use_phase_blending or cfg.use_wave_rendering,
                                 use_edge_aware=cfg.use_edge_aware, edge_scale_factor=cfg.edge_scale_factor,
                                 edge_opacity_boost=cfg.edge_opacity_boost).to(device)
    dp.broadcast_parameters(model)
    train_res = hfts.get_effective_train_resolution(cfg.image_size) if hfts is not None else cfg.image_size
    renderer, camera = _make_renderer(renderer_factory, cfg, device, train_res)
    # host-side draws that every rank must make identically (the pose is shared by the global batch; the K sampled
    # indices differ per rank like any other per-shard quantity)
    pose_rng = np.random.RandomState(cfg.seed + 7919)
    sample_gen = torch.Generator(device=device).manual_seed(cfg.seed * 104729 + dp.rank) if device.type == "cuda" else None
    optimizer = make_optimizer(model, cfg)
    scheduler = CosineAnnealingLR(optimizer, T_max=cfg.epochs)
    start_epoch = 0
    if resume:
        ck = torch.load(resume, map_location=device)
        model.load_state_dict(ck["model_state_dict"])
        optimizer.load_state_dict(ck["optimizer_state_dict"])
        start_epoch = ck["epoch"] + 1
    data, n_items = make_dataset(cfg, log if dp.rank == 0 else (lambda *a: None))
    # metrics beyond the losses: composited Gaussian-pixels counted on the device by the renderer,
    # the rasterizer's stage timers sampled on the logged steps
    pair_counter = None
    stage_timers = None
    if device.type == "cuda" and hasattr(renderer, "pair_counter"):
        pair_counter = torch.zeros(3, dtype=torch.int64, device=device)  # [pairs, tile duplicates, Gaussians]
        renderer.pair_counter = pair_counter
        from . import _binding as stage_timers
    history: Dict[str, list] = {}
    epoch_history = []
    # one captured graph for the whole run: needs a step whose host side is the same every time (no per-batch pose, no
    # per-epoch Gaussian count / K-subset schedule)
    use_graph = bool(cfg.hip_graph) and device.type == "cuda" and hfts is None and not cfg.multi_pose_augmentation
    if cfg.hip_graph and not use_graph and dp.rank == 0:
        log("  --hip_graph ignored: the step changes from batch to batch (HFTS schedule / per-batch pose) or runs on the CPU")
    graphed = None
    for epoch in range(start_epoch, cfg.epochs):
        model.train()
        batches = epoch_batches(n_items, cfg.batch_size, dp.world, cfg.seed, epoch)
        shards = []
        for gb in batches:
            lo, hi = dp.shard(len(gb))  # image-wise shard of the global batch
            shards.append(gb[lo:hi])
        acc: Dict[str, torch.Tensor] = {}
        n_ok = torch.zeros((), device=device)
        if pair_counter is not None:
            pair_counter.zero_()
            stage_timers.stage_timing_read()
        t0 = time.perf_counter()
        for bi, batch in enumerate(BatchPrefetcher(data, shards, device, cfg.num_workers)):
            logged = bi % cfg.log_interval == 0
            if use_graph and graphed is None and len(shards[bi]) * dp.world == cfg.batch_size:
                graphed = GraphedTrainStep(model, renderer, camera, optimizer, cfg, dp, batch, epoch=epoch, train_res=train_res)
            if graphed is not None and graphed.matches(batch):
                res = graphed(batch)  # (stage timers are event pairs recorded at launch time: not inside a replayed graph)
            else:
                if stage_timers is not None and logged:
                    stage_timers.stage_timing_enable(True)
                res = train_step(model, renderer, camera, batch, optimizer, cfg, dp, hfts=hfts, epoch=epoch,
                                 train_res=train_res, pose_rng=pose_rng, sample_gen=sample_gen)
                if stage_timers is not None and logged:
                    stage_timers.stage_timing_enable(False)
            ok = 1.0 - res.skipped
            n_ok = n_ok + ok
            for k, v in res.terms.items():  # device-side sums over the batches that were not skipped
                acc[k] = acc.get(k, 0.0) + torch.where(ok > 0, v.float(), torch.zeros_like(v, dtype=torch.float32))
            if logged:  # the only host round trip inside the epoch: once per log_interval (TGD:1275-1281)
                ld = res.to_host()
                if dp.rank == 0:
                    if ld is None:
                        log(f"  Warning: NaN/Inf loss at batch {bi}, skipping")
                    else:
                        log(f"  Batch {bi}/{len(shards)} | Loss: {ld['total']:.4f} | RGB: {ld['rgb']:.4f}")
        scheduler.step()
        keys = sorted(acc)
        extra = [pair_counter[i].double() for i in range(3)] if pair_counter is not None else []
        vals = torch.stack([acc[k].double() for k in keys] + [n_ok.double()] + extra)
        if dp.enabled and pair_counter is not None:
            import torch.distributed as tdist  # whole-job pair / duplicate counts (the losses are this rank's shard, as before)
            tdist.all_reduce(vals[-3:], op=tdist.ReduceOp.SUM)
        vals = vals.tolist()  # epoch end: one transfer (also the sync that closes the epoch's clock)
        elapsed = time.perf_counter() - t0
        nb = vals[len(keys)]
        losses = {k: v / nb for k, v in zip(keys, vals)} if nb > 0 else {}
        epoch_history.append(losses)
        steps = max(len(shards), 1)
        metrics = {"step_ms": elapsed / steps * 1e3, "skipped_batches": len(shards) - int(nb)}
        if pair_counter is not None:
            pairs, dups, gauss = vals[-3], vals[-2], vals[-1]
            metrics["pairs_per_s"] = pairs / max(elapsed, 1e-9)
            # ALGORITHMIC HBM bytes of the rasterizer per second, whole job, forward + backward: the per-stage figures of
            # DESIGN.md section 4 / bench.py -- 304 B per Gaussian (projection 112, depth sort 72, lists 8, projection
            # adjoint 112), 188 B per tile duplicate (lists 4, record gather 52 + 52, gradient row 40 + 40) and 72 B per
            # pixel (state, outputs, upstream gradients) -- over the epoch's wall time (which includes the decoder)
            n_img = sum(len(sh) for sh in shards) * dp.world
            metrics["hbm_gbs_algorithmic"] = (304.0 * gauss + 188.0 * dups + 72.0 * n_img * train_res * train_res) / max(elapsed, 1e-9) / 1e9
            st = stage_timers.stage_timing_read()
            metrics["stage_ms"] = {k: v[0] / v[1] for k, v in st.items() if v[1]}
        for k, v in list(losses.items()) + list(metrics.items()):
            if isinstance(v, dict):
                for kk, vv in v.items():
                    history.setdefault(k, {}).setdefault(kk, []).append(vv)
            else:
                history.setdefault(k, []).append(v)
        if dp.rank == 0:
            log(f"Epoch {epoch + 1}/{cfg.epochs} | Time: {elapsed:.1f}s | {metrics['step_ms']:.2f} ms/step | "
                + " ".join(f"{k}={v:.4f}" for k, v in losses.items()))
            if (epoch + 1) % cfg.save_interval == 0 or epoch + 1 == cfg.epochs:
                save_checkpoint(model, optimizer, epoch, losses, cfg)
    if dp.rank == 0 and history:
        save_training_history(history, cfg)
    return model, epoch_history


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train the Gaussian decoder through the HIP rasterizer")
    c = TrainingConfig()
    ap.add_argument("--experiment", type=int, default=c.experiment)
    ap.add_argument("--data_dir", default=c.data_dir)
    ap.add_argument("--output_dir", default=c.output_dir)
    ap.add_argument("--batch_size", type=int, default=c.batch_size)
    ap.add_argument("--epochs", type=int, default=c.epochs)
    ap.add_argument("--lr", type=float, default=c.lr)
    ap.add_argument("--image_size", type=int, default=c.image_size)
    ap.add_argument("--gaussians_per_patch", type=int, default=c.gaussians_per_patch)
    ap.add_argument("--max_images", type=int, default=None)
    ap.add_argument("--use_fresnel_zones", type=int, nargs="?", const=8, default=0,
                    help="quantise depth into N zones (TGD flag; bare flag = 8)")
    ap.add_argument("--num_fresnel_zones", type=int, default=c.num_fresnel_zones)
    ap.add_argument("--use_phase_blending", action="store_true")
    ap.add_argument("--phase_amplitude", type=float, default=c.phase_amplitude)
    ap.add_argument("--use_edge_aware", action="store_true", help="smaller Gaussians at depth edges (TGD:1455)")
    ap.add_argument("--edge_scale_factor", type=float, default=c.edge_scale_factor, help="TGD:1461")
    ap.add_argument("--train_resolution", type=int, default=None, help="HFTS render resolution (TGD:1524)")
    ap.add_argument("--progressive_schedule", action="store_true", help="HFTS progressive Gaussian growing (TGD:1526)")
    ap.add_argument("--stochastic_k", type=int, default=None, help="HFTS: render K importance-sampled Gaussians (TGD:1528)")
    ap.add_argument("--fast_mode", action="store_true", help="HFTS preset: 64x64, progressive, K=256 (TGD:1530)")
    ap.add_argument("--multi_pose_augmentation", action="store_true", help="random orbit pose per batch (TGD:1536)")
    ap.add_argument("--pose_range_elevation", type=float, nargs=2, default=[-30, 45], help="degrees (TGD:1538)")
    ap.add_argument("--pose_range_azimuth", type=float, nargs=2, default=[0, 360], help="degrees (TGD:1540)")
    ap.add_argument("--frontal_prob", type=float, default=c.frontal_prob, help="TGD:1542")
    ap.add_argument("--use_pose_encoding", action="store_true", help="TGD:1544 (pose-dependent decoder output)")
    ap.add_argument("--use_wave_rendering", action="store_true", help="WaveFieldRenderer (TGD:1469)")
    ap.add_argument("--use_qsr", action="store_true", help="macro flag: implies --use_wave_rendering (TGD:1550-1553)")
    ap.add_argument("--wave_equation_weight", type=float, default=c.wave_equation_weight, help="TGD:1483")
    ap.add_argument("--wavelength", type=float, default=c.wavelength)
    ap.add_argument("--use_phase_retrieval_loss", action="store_true", help="TGD:1494 (--use_qsr implies it, TGD:1553)")
    ap.add_argument("--phase_retrieval_weight", type=float, default=c.phase_retrieval_weight)
    ap.add_argument("--use_frequency_loss", action="store_true", help="TGD:1498")
    ap.add_argument("--frequency_loss_weight", type=float, default=c.frequency_loss_weight)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--hip_graph", action="store_true", help="replay the whole training step from one captured HIP graph")
    ap.add_argument("--renderer", default="hip", choices=["hip"],
                    help="only the HIP rasterizer ships; there is no CPU fallback")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    if a.experiment != 2:
        raise SystemExit("only --experiment 2 (direct patch decoder) is on this repo's hot path")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("fresnel_amd.train needs a GPU: the HIP rasterizer has no CPU fallback")
    torch.cuda.set_device(local_rank)
    cfg = TrainingConfig(experiment=a.experiment, data_dir=a.data_dir, output_dir=a.output_dir,
                         batch_size=a.batch_size, epochs=a.epochs, lr=a.lr, image_size=a.image_size,
                         gaussians_per_patch=a.gaussians_per_patch, max_images=a.max_images,
                         use_fresnel_zones=bool(a.use_fresnel_zones),
                         num_fresnel_zones=a.use_fresnel_zones or a.num_fresnel_zones,
                         use_phase_blending=a.use_phase_blending, phase_amplitude=a.phase_amplitude,
                         use_edge_aware=a.use_edge_aware, edge_scale_factor=a.edge_scale_factor,
                         multi_pose_augmentation=a.multi_pose_augmentation,
                         pose_range_elevation=tuple(a.pose_range_elevation), pose_range_azimuth=tuple(a.pose_range_azimuth),
                         frontal_prob=a.frontal_prob, use_pose_encoding=a.use_pose_encoding,
                         use_wave_rendering=a.use_wave_rendering or a.use_qsr,
                         wave_equation_weight=a.wave_equation_weight, wavelength=a.wavelength,
                         use_phase_retrieval_loss=a.use_phase_retrieval_loss or a.use_qsr,
                         phase_retrieval_weight=a.phase_retrieval_weight,
                         use_frequency_loss=a.use_frequency_loss, frequency_loss_weight=a.frequency_loss_weight,
                         device=f"cuda:{local_rank}", seed=a.seed, hip_graph=a.hip_graph)
    hfts = HFTSConfig(train_resolution=a.train_resolution, progressive_schedule=a.progressive_schedule,
                      stochastic_k=a.stochastic_k, fast_mode=a.fast_mode)
    dp = DPContext(device=torch.device(cfg.device))
    try:
        run_training(cfg, dp, resume=a.resume, hfts=hfts if hfts.enabled else None)
    finally:
        dp.shutdown()


if __name__ == "__main__":
    main()
