#!/usr/bin/env python
"""Edge cases of lib/aligning.py's two similarity estimators, produced by IMPORTING the reference (build container only):
    python tests/golden/gen_transform_edge_golden.py        -> tests/golden/transform_edge.npz
  * estimateSimilarityTransform (:17-32) on a target UNRELATED to the source: no 5-point model explains 10 % of the points, the
    reference prints its warning and returns 4 x None.  Inputs, the replayed randint stream and the reference's answer are stored.
  * estimateSimilarityUmeyama (:580-622) on float64 inputs that are NOT float32-representable: the product path carries points
    as float32 across the C ABI (pose/aligning.py), so these goldens bound that quantisation against the reference's float64 result.
The oracle is asserted bit-equal to the reference on the way (same shims as gen_pose_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_pose_golden import import_reference, quiet, rand_rot  # noqa: E402


def main():
    _d3, _pose, aligning = import_reference()
    from oracle import pose_oracle as PO
    out = {}
    rng = np.random.RandomState(300)
    n = 300
    src, tgt = rng.uniform(-20, 20, (n, 3)), rng.uniform(-20, 20, (n, 3))
    rs = np.random.RandomState(11)
    draws = np.stack([rs.randint(n, size=5) for _ in range(100)])
    np.random.seed(11)
    with quiet():
        ref = aligning.estimateSimilarityTransform(src, tgt)
    with quiet():
        orc = PO.estimateSimilarityTransform(src, tgt, draws)
    assert all(r is None for r in ref) and all(o is None for o in orc), (ref, orc)
    out.update(none_src=src, none_tgt=tgt, none_draws=draws.astype(np.int32), none_is_none=np.asarray(1))
    for i, m in enumerate((40, 500)):
        s = rng.uniform(-1, 1, (m, 3)) * np.pi / 3.0                      # float64 values with full mantissas
        R, sc, t = rand_rot(rng), rng.uniform(0.5, 2.0), rng.uniform(-1, 1, 3)
        g = sc * s @ R.T + t + rng.randn(m, 3) * 1e-3
        S, Rot, T, Out = aligning.estimateSimilarityUmeyama(s.transpose(), g.transpose())
        S2, R2, T2, O2 = PO.estimateSimilarityUmeyama(s.transpose(), g.transpose())
        assert np.array_equal(S, S2) and np.array_equal(Rot, R2) and np.array_equal(T, T2) and np.array_equal(Out, O2)
        assert not np.array_equal(s, s.astype(np.float32).astype(np.float64))
        out.update({f"f64_src{i}": s, f"f64_tgt{i}": g, f"f64_S{i}": S, f"f64_R{i}": Rot, f"f64_T{i}": T, f"f64_Out{i}": Out})
    out["f64_cases"] = np.asarray(2)
    np.savez_compressed(os.path.join(HERE, "transform_edge.npz"), **out)
    print("transform_edge.npz", os.path.getsize(os.path.join(HERE, "transform_edge.npz")))


if __name__ == "__main__":
    main()
