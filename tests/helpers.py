"""Shared test inputs for the network-half operators."""
import numpy as np


def cloud(rng, b, n, kind="uniform"):
    """Point clouds in [-1,1]^3.  kinds: uniform floats; grid = coordinates on a 2^-8 lattice (all
    squared-distance arithmetic exact in float32 -> every FMA-contraction pattern agrees, and exact
    distance ties are common); tiled = a small cloud repeated to n points (the reference loader
    tiles small clouds, lib/dataset.py:290-317 -> duplicated points)."""
    if kind == "uniform":
        return rng.uniform(-1, 1, (b, n, 3)).astype(np.float32)
    if kind == "grid":
        return (rng.randint(-256, 257, (b, n, 3)) / 256.0).astype(np.float32)
    if kind == "coarse":   # very coarse lattice: massive ties
        return (rng.randint(-8, 9, (b, n, 3)) / 8.0).astype(np.float32)
    if kind == "tiled":
        base = rng.uniform(-1, 1, (b, max(1, n // 3), 3)).astype(np.float32)
        reps = -(-n // base.shape[1])
        return np.tile(base, (1, reps, 1))[:, :n].copy()
    raise ValueError(kind)


# ---- hand-built weights for the coupled (production) data-flow tests: the builder lives in the package so that bench.py
# --pose-inputs network can use it too
from articulated_pose_amd.synthetic import passthrough_pose_problem  # noqa: E402,F401
