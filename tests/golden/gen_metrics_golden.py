#!/usr/bin/env python
"""Golden vectors of the evaluation metrics, produced by IMPORTING the reference's lib/d3_utils.py from /root/reference
(build container only).  Writes tests/golden/metrics.npz and asserts that oracle/metrics_oracle.py reproduces the reference
exactly on the same inputs (that is what pins the oracle).

    python tests/golden/gen_metrics_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path[:0] = ["/root/reference", "/root/reference/lib"]
import matplotlib  # noqa: E402

matplotlib.use("Agg")
from lib import d3_utils as ref  # noqa: E402

from oracle import metrics_oracle as orc  # noqa: E402


def rand_rot(rng):
    q = rng.randn(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def main():
    rng = np.random.RandomState(20240928)
    b1, b2, iou, cnt = [], [], [], []
    for case in range(14):
        s1 = rng.uniform(0.2, 1.0, 3)
        base = ref.get_3d_bbox(s1, shift=np.array([0.5, 0.5, 0.5])).transpose()
        assert np.array_equal(base, orc.get_3d_bbox(s1, shift=np.array([0.5, 0.5, 0.5])).transpose())
        R1, t1 = rand_rot(rng), rng.uniform(-0.3, 0.3, 3)
        if case == 0:            # identical boxes
            R2, t2, s2 = R1, t1, s1
        elif case == 1:          # disjoint boxes
            R2, t2, s2 = rand_rot(rng), t1 + 5.0, rng.uniform(0.2, 1.0, 3)
        elif case == 2:          # axis-aligned, nested
            R1 = R2 = np.eye(3); t2 = t1; s2 = s1 * 0.5
        else:                    # near-by pose, as a prediction vs ground truth
            d = rand_rot(rng)
            a = rng.uniform(0.0, 0.3)
            R2 = (np.eye(3) * (1 - a) + d * a)
            u, _, vt = np.linalg.svd(R2)
            R2 = (u @ vt) @ R1
            t2, s2 = t1 + rng.uniform(-0.1, 0.1, 3), s1 * rng.uniform(0.8, 1.2, 3)
        bb1 = np.dot(base, R1.T) + t1
        bb2 = np.dot(ref.get_3d_bbox(s2, shift=np.array([0.5, 0.5, 0.5])).transpose(), R2.T) + t2
        v = ref.iou_3d(bb1, bb2)
        ov, oi, ou = orc.iou_3d(bb1, bb2, return_counts=True)
        assert v == ov, (case, v, ov)
        b1.append(bb1); b2.append(bb2); iou.append(float(v)); cnt.append([oi, ou])
    small = [ref.iou_3d(x, y, nres=17) for x, y in zip(b1, b2)]
    assert small == [orc.iou_3d(x, y, nres=17) for x, y in zip(b1, b2)]
    # scalar metrics
    R = np.stack([rand_rot(rng) for _ in range(16)]); Q = np.stack([rand_rot(rng) for _ in range(16)])
    rd = np.array([ref.rot_diff_degree(a, b) for a, b in zip(R, Q)])
    assert np.array_equal(rd, np.array([orc.rot_diff_degree(a, b) for a, b in zip(R, Q)]))
    v1, v2 = rng.randn(16, 3), rng.randn(16, 3)
    ad = np.array([ref.axis_diff_degree(a, b) for a, b in zip(v1, v2)])
    assert np.array_equal(ad, np.array([orc.axis_diff_degree(a, b) for a, b in zip(v1, v2)]))
    p1, p2 = rng.randn(16, 3), rng.randn(16, 3)
    ld = np.array([ref.dist_between_3d_lines(a, b, c, d) for a, b, c, d in zip(p1, v1, p2, v2)])
    assert np.array_equal(ld, np.array([orc.dist_between_3d_lines(a, b, c, d) for a, b, c, d in zip(p1, v1, p2, v2)]))
    np.savez(os.path.join(HERE, "metrics.npz"), bbox1=np.stack(b1), bbox2=np.stack(b2), iou=np.array(iou), counts=np.array(cnt),
             iou_nres17=np.array(small, np.float64), R=R, Q=Q, rot_diff_degree=rd, v1=v1, v2=v2, axis_diff_degree=ad, p1=p1, p2=p2,
             line_dist=ld)
    print("wrote metrics.npz:", len(iou), "box pairs; iou", np.round(iou, 4))


if __name__ == "__main__":
    main()
