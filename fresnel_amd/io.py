"""On-disk formats on either side of the rasterizer.

* N x 14 float32 Gaussian file = position(3) scale(3) rotation(4, wxyz) color(3) opacity(1): same functions and dict
  keys as load_gaussians_from_binary / save_gaussians_to_binary (DR:1461-1497; C++ GaussianCloud binary I/O
  src/core/renderer/renderer.cpp:557-647).
* 3DGS-style PLY of the C++ viewer (GaussianCloud::save_ply / load_ply, src/core/renderer/renderer.cpp:649-793):
  ASCII header, binary_little_endian, 14 float properties x y z scale_0..2 (LOG scale) rot_0..3 (wxyz)
  f_dc_0..2 ((colour - 0.5) / C0) opacity (inverse sigmoid).
* Training caches next to the images (ImageDataset.__getitem__, TGD:603-630): `<name>_dinov2[_base|_large].bin` =
  37 x 37 x feature_dim float32 (patch-major), `<name>_depth.bin` = S x S float32 in [0, 1].
"""
import os

import numpy as np
import torch

FLOATS_PER_GAUSSIAN = 14


def load_gaussians_from_binary(path: str) -> dict:
    data = np.fromfile(path, dtype=np.float32)
    n = len(data) // FLOATS_PER_GAUSSIAN
    data = data[:n * FLOATS_PER_GAUSSIAN].reshape(n, FLOATS_PER_GAUSSIAN)
    return {
        "positions": torch.from_numpy(data[:, 0:3].copy()),
        "scales": torch.from_numpy(data[:, 3:6].copy()),
        "rotations": torch.from_numpy(data[:, 6:10].copy()),
        "colors": torch.from_numpy(data[:, 10:13].copy()),
        "opacities": torch.from_numpy(data[:, 13].copy()),
    }


def save_gaussians_to_binary(path: str, gaussians: dict):
    n = gaussians["positions"].shape[0]
    data = np.zeros((n, FLOATS_PER_GAUSSIAN), dtype=np.float32)
    data[:, 0:3] = gaussians["positions"].detach().cpu().numpy()
    data[:, 3:6] = gaussians["scales"].detach().cpu().numpy()
    data[:, 6:10] = gaussians["rotations"].detach().cpu().numpy()
    data[:, 10:13] = gaussians["colors"].detach().cpu().numpy()
    data[:, 13] = gaussians["opacities"].detach().cpu().numpy()
    data.tofile(path)


# ---- 3DGS-style PLY (src/core/renderer/renderer.cpp:649-793) -------------------------------------------------------
SH_C0 = np.float32(0.28209479177387814)  # SH basis of the constant term (renderer.cpp:693)
PLY_PROPERTIES = ["x", "y", "z", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3",
                  "f_dc_0", "f_dc_1", "f_dc_2", "opacity"]


def save_ply(path: str, gaussians: dict):
    """GaussianCloud::save_ply (renderer.cpp:649-712): log(max(scale, 1e-7)), f_dc = (colour - 0.5) / C0,
    opacity = log(o / max(1 - o, 1e-7)), all float32 arithmetic."""
    n = gaussians["positions"].shape[0]
    f32 = lambda k: gaussians[k].detach().cpu().numpy().astype(np.float32)
    data = np.zeros((n, 14), dtype="<f4")
    data[:, 0:3] = f32("positions")
    data[:, 3:6] = np.log(np.maximum(f32("scales"), np.float32(1e-7)))
    data[:, 6:10] = f32("rotations")
    data[:, 10:13] = (f32("colors") - np.float32(0.5)) / SH_C0
    o = f32("opacities").reshape(n)
    with np.errstate(divide="ignore"):
        data[:, 13] = np.log(o / np.maximum(np.float32(1.0) - o, np.float32(1e-7)))
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join("property float %s\n" % p for p in PLY_PROPERTIES) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(data.tobytes())


def load_ply(path: str) -> dict:
    """GaussianCloud::load_ply (renderer.cpp:714-793): reads `element vertex N`, skips to end_header (CR tolerated),
    then N x 14 little-endian floats in the property order above; scale = exp, colour = clamp(f_dc C0 + 0.5, 0, 1),
    opacity = sigmoid.  Raises ValueError for a bad header / truncated body (the C++ returns false)."""
    with open(path, "rb") as f:
        n, done = 0, False
        while True:
            line = f.readline()
            if not line:
                break
            text = line.decode("ascii", errors="replace").rstrip("\n").rstrip("\r")
            if "element vertex" in text:
                parts = text.split()
                n = int(parts[2]) if len(parts) > 2 else 0
            elif text == "end_header":
                done = True
                break
        if not done or n == 0:
            raise ValueError(f"{path}: invalid PLY header or no vertices")
        raw = f.read(n * 14 * 4)
    if len(raw) < n * 14 * 4:
        raise ValueError(f"{path}: failed reading Gaussian {len(raw) // 56} of {n}")
    v = np.frombuffer(raw, dtype="<f4").reshape(n, 14).astype(np.float32)
    with np.errstate(over="ignore"):
        opacity = np.float32(1.0) / (np.float32(1.0) + np.exp(-v[:, 13]))
    return {
        "positions": torch.from_numpy(v[:, 0:3].copy()),
        "scales": torch.from_numpy(np.exp(v[:, 3:6])),
        "rotations": torch.from_numpy(v[:, 6:10].copy()),
        "colors": torch.from_numpy(np.clip(v[:, 10:13] * SH_C0 + np.float32(0.5), 0.0, 1.0).astype(np.float32)),
        "opacities": torch.from_numpy(opacity.astype(np.float32)),
    }


def load_gaussians(path: str) -> dict:
    """.ply or .bin by extension, like the viewer (src/viewer/viewer.cpp:266-279)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".ply":
        return load_ply(path)
    if ext == ".bin":
        return load_gaussians_from_binary(path)
    raise ValueError(f"unsupported Gaussian file {path!r}: supported formats .ply, .bin")


# ---- training caches (TGD:603-630) ----------------------------------------------------------------------------------
FEATURE_GRID = 37


def feature_cache_suffix(feature_dim: int) -> str:
    """`_dinov2.bin` for the small model (384), `_dinov2_base.bin` / `_dinov2_large.bin` for 768 / 1024 (TGD:552-555)."""
    size = {384: "small", 768: "base", 1024: "large"}.get(feature_dim, "small")
    return "_dinov2.bin" if size == "small" else f"_dinov2_{size}.bin"


def load_feature_cache(path: str, feature_dim: int = 384) -> torch.Tensor:
    """`<name>_dinov2*.bin` -> (feature_dim, 37, 37): the file is patch-major 37 x 37 x feature_dim (TGD:611-614)."""
    f = np.fromfile(path, dtype=np.float32)
    return torch.from_numpy(f.reshape(FEATURE_GRID, FEATURE_GRID, feature_dim).transpose(2, 0, 1).copy())


def load_depth_cache(path: str, image_size: int) -> torch.Tensor:
    """`<name>_depth.bin` -> (1, image_size, image_size).  The cache is a square float32 map (256 x 256 from the
    preprocessing); a different target size goes through the reference's 8-bit PIL bilinear resize (TGD:620-627:
    (depth * 255) -> uint8 'L' image -> resize -> / 255), reproduced here with PIL."""
    d = np.fromfile(path, dtype=np.float32)
    s = int(np.sqrt(len(d)))
    d = d.reshape(s, s)
    if s != image_size:
        from PIL import Image
        img = Image.fromarray((d * 255).astype(np.uint8), mode="L")
        img = img.resize((image_size, image_size), Image.Resampling.BILINEAR)
        d = np.array(img, dtype=np.float32) / 255.0
    return torch.from_numpy(d.copy()).unsqueeze(0)
