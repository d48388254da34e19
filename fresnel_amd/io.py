"""On-disk Gaussian format shared by the reference's Python and C++ sides:
N x 14 float32 = position(3) scale(3) rotation(4, wxyz) color(3) opacity(1).
Same functions and dict keys as load_gaussians_from_binary / save_gaussians_to_binary
(DR:1461-1497; C++ GaussianCloud binary I/O src/core/renderer/renderer.cpp:557-647)."""
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
