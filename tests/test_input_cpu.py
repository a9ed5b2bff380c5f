"""CPU: oracle/input_oracle.py against the golden record produced by the reference's own
Dataset.create_unit_data_from_hdf5 (tests/golden/gen_input_golden.py), bit for bit."""
import os

import numpy as np
import pytest

from oracle import input_oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "input_sampling.npz")
IN_KEYS = ("parts_pts", "parts_cls", "nocs_p", "nocs_g", "offset_heatmap", "offset_unitvec", "joint_orient", "joint_cls")
OUT_KEYS = ("P", "cls_gt", "mask_array", "nocs_gt", "nocs_gt_g", "heatmap_gt", "unitvec_gt", "orient_gt", "joint_cls_gt", "joint_cls_mask")


def load_case(z, tag):
    sizes = z[f"{tag}_sizes"]
    cuts = np.cumsum(sizes)[:-1]
    parts = {k: np.split(z[f"{tag}_in_{k}"], cuts) for k in IN_KEYS}
    want = {k: z[f"{tag}_out_{k}"] for k in OUT_KEYS}
    return parts, int(z[f"{tag}_num_points"]), float(z[f"{tag}_norm_factor"]), len(sizes), z[f"{tag}_perm"], want


def cases():
    with np.load(G) as z:
        return [str(c) for c in z["cases"]]


@pytest.mark.parametrize("tag", cases())
def test_input_oracle_equals_reference_record(tag):
    with np.load(G) as z:
        parts, N, nf, K, perm, want = load_case(z, tag)
    got = input_oracle.create_unit_data(parts, N, nf, K, perm=perm)
    for k in OUT_KEYS:
        assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
    assert got["P"].shape == (N, 3) and got["mask_array"].sum() == N


def test_tiling_is_a_modulo():
    """The product never materialises the tiled cloud: tiled[t] == raw[t % n_raw]."""
    raw = np.arange(7)
    tiled = np.concatenate([raw] * 3)
    assert np.array_equal(tiled, raw[np.arange(21) % 7])
