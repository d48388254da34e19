"""Shrinks the committed fixtures without changing a value (SURVEY 8c asked for < 300 KB each): arrays that
are REGENERABLE bit for bit are dropped and rebuilt by tests/helpers.load_golden on access --
  * gI / gD: drawn by make_goldens.upstream(seed_up, H, W) from numpy's frozen legacy RandomState (checked here before dropping);
  * depth_order of at most 65 536 Gaussians: stored as uint16 (cast back to int32 by the loader).
make_goldens.py calls slim() on every fixture it writes; run this file to re-slim fixtures in place:  python tests/golden/slim_goldens.py"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _upstream(seed, H, W):
    rs = np.random.RandomState(seed)
    gI = rs.standard_normal((3, H, W)).astype(np.float32)
    gD = (rs.standard_normal((H, W)) * 0.1).astype(np.float32)
    return gI, gD


def slim(path):
    g = np.load(path)
    rec = {k: g[k] for k in g.files}
    before = os.path.getsize(path)
    if "gI" in rec and "seed_up" in rec and rec["gI"].ndim == 3:
        _, H, W = rec["gI"].shape
        gI, gD = _upstream(int(rec["seed_up"]), H, W)
        if np.array_equal(gI, rec["gI"]) and ("gD" not in rec or np.array_equal(gD, rec["gD"])):
            rec["upstream_shape"] = np.array([H, W], np.int32)
            rec["upstream_has_gD"] = np.int32(1 if "gD" in rec else 0)
            rec.pop("gI")
            rec.pop("gD", None)
    if "depth_order" in rec and rec["depth_order"].dtype == np.int32 and rec["depth_order"].size <= 65536 and rec["depth_order"].min() >= 0:
        rec["depth_order"] = rec["depth_order"].astype(np.uint16)
    np.savez_compressed(path, **rec)
    return before, os.path.getsize(path)


if __name__ == "__main__":
    for f in sorted(glob.glob(os.path.join(HERE, "*.npz")) if len(sys.argv) < 2 else sys.argv[1:]):
        b, a = slim(f)
        print(f"{os.path.basename(f):40s} {b // 1024:5d} KB -> {a // 1024:5d} KB")
