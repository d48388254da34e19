"""Dataset side of the training harness: this repo's counterpart of the reference's ImageDataset
(scripts/training/train_gaussian_decoder.py:525-675) for the tensors the rasterizer path consumes.

    ImageDataset(data_dir, image_size, feature_cache_dir=None, max_images=None, feature_dim=384)[i] ->
        {'image' (3,S,S) in [0,1], 'features' (feature_dim,37,37), 'depth' (1,S,S), 'has_saag', 'saag_*', 'name'}

Same file discovery (jpg / jpeg / png / webp, both cases, sorted, max_images prefix), the same cache file names and
layouts (fresnel_amd/io.py), the same fall-backs (zero features / zero depth / empty SAAG when a cache is missing).
Colour-jitter augmentation and the VLM density maps are not part of the rasterizer path and are left out.
"""
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

from . import io as fio


class ImageDataset:
    def __init__(self, data_dir: str, image_size: int = 256, feature_cache_dir: Optional[str] = None,
                 max_images: Optional[int] = None, feature_dim: int = 384):
        self.data_dir = Path(data_dir)
        self.image_size = image_size
        self.feature_cache_dir = Path(feature_cache_dir) if feature_cache_dir else self.data_dir / "features"
        self.feature_dim = feature_dim
        self.feature_suffix = fio.feature_cache_suffix(feature_dim)
        paths: List[Path] = []
        for ext in ["*.jpg", "*.jpeg", "*.png", "*.webp"]:
            paths.extend(self.data_dir.glob(ext))
            paths.extend(self.data_dir.glob(ext.upper()))
        self.image_paths = sorted(paths)
        if max_images is not None and len(self.image_paths) > max_images:
            self.image_paths = self.image_paths[:max_images]

    def __len__(self) -> int:
        return len(self.image_paths)

    def _step_tensors(self, idx: int):
        """image, features (C,37,37), depth of item `idx` -- the three tensors a training step consumes."""
        from PIL import Image
        path = self.image_paths[idx]
        name = path.stem
        S = self.image_size
        img = Image.open(path).convert("RGB").resize((S, S), Image.Resampling.LANCZOS)
        image = (torch.from_numpy(np.array(img)).float() / 255.0).permute(2, 0, 1)
        fpath = self.feature_cache_dir / f"{name}{self.feature_suffix}"
        dpath = self.feature_cache_dir / f"{name}_depth.bin"
        features = (fio.load_feature_cache(str(fpath), self.feature_dim) if fpath.exists()
                    else torch.zeros(self.feature_dim, fio.FEATURE_GRID, fio.FEATURE_GRID))
        depth = fio.load_depth_cache(str(dpath), S) if dpath.exists() else torch.zeros(1, S, S)
        return name, image, features, depth

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        name, image, features, depth = self._step_tensors(idx)
        spath = self.feature_cache_dir / f"{name}_saag.bin"
        item = {"image": image, "features": features, "depth": depth, "has_saag": spath.exists(), "name": name}
        if spath.exists():
            saag = fio.load_gaussians_from_binary(str(spath))
            for k in ("positions", "scales", "rotations", "colors", "opacities"):
                item["saag_" + k] = saag[k].float()
        else:
            item.update(saag_positions=torch.zeros(0, 3), saag_scales=torch.zeros(0, 3), saag_rotations=torch.zeros(0, 4),
                        saag_colors=torch.zeros(0, 3), saag_opacities=torch.zeros(0))
        return item

    def host_item(self, idx: int):
        """(image (3,S,S), features (37,37,C) patch-major as the decoder takes them, depth (1,S,S)) on the host: what the
        training loop's prefetch threads load ahead of the step (fresnel_amd/train.py BatchPrefetcher)."""
        _, image, features, depth = self._step_tensors(idx)  # (not self[idx]: the SAAG binaries are not part of a step)
        return image, features.permute(1, 2, 0).contiguous(), depth

    def batch(self, indices, device):
        """(images (B,3,S,S), features (B,37,37,C) patch-major as the decoder takes them, depth (B,1,S,S))."""
        items = [self[i] for i in indices]
        images = torch.stack([it["image"] for it in items]).to(device)
        feats = torch.stack([it["features"].permute(1, 2, 0) for it in items]).to(device)
        depth = torch.stack([it["depth"] for it in items]).to(device)
        return images, feats, depth
