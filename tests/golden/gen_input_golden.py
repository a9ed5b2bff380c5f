#!/usr/bin/env python
"""Golden vectors of the input sampling step, produced by RUNNING the reference's own
lib/dataset.py::Dataset.create_unit_data_from_hdf5 (:251-432) from /root/reference (build container only).

The method parses an .h5 frame through create_data_shape2motion and then tiles / permutes / scales.  No dataset is
available offline, so the parser is replaced by a stub that hands the method seeded synthetic per-part arrays (the
method's own lines 262-432 -- concatenation, tiling, permutation, norm_factor scaling, masks, record assembly -- run
unmodified).  Import shims as in gen_pose_golden.py (h5py, cv2, ... absent here).  Stores the per-part inputs, the
permutation numpy drew (replayed from the same seed), and the reference's record; asserts oracle/input_oracle.py equal.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"


def import_dataset():
    sys.path[:0] = [REF, os.path.join(REF, "lib"), os.path.join(REF, "evaluation")]
    import matplotlib
    matplotlib.use("Agg")
    for name in ("h5py", "cv2", "trimesh", "descartes", "tensorflow", "keras"):
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["descartes"].PolygonPatch = object
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        import dataset
    finally:
        os.chdir(cwd)
    return dataset


def synthetic_parts(rng, sizes):
    K = len(sizes)
    parts = dict(parts_pts=[], parts_cls=[], nocs_p=[], nocs_g=[], offset_heatmap=[], offset_unitvec=[], joint_orient=[], joint_cls=[])
    for j, n in enumerate(sizes):
        parts["parts_pts"].append(rng.uniform(-1, 1, (n, 3)).astype(np.float32))
        parts["parts_cls"].append(np.full(n, j, np.float32))
        parts["nocs_p"].append(rng.uniform(0, 1, (n, 3)).astype(np.float32))
        parts["nocs_g"].append(rng.uniform(0, 1, (n, 3)).astype(np.float32))
        parts["offset_heatmap"].append(rng.uniform(0, 1, n).astype(np.float32))
        parts["offset_unitvec"].append(rng.randn(n, 3).astype(np.float32))
        parts["joint_orient"].append(rng.randn(n, 3).astype(np.float32))
        parts["joint_cls"].append(rng.randint(0, K, n).astype(np.float32))
    return parts


def main():
    dataset = import_dataset()
    from oracle import input_oracle
    ds = object.__new__(dataset.Dataset)                       # no __init__: it walks the on-disk split files
    ds.name_dset = "shape2motion"
    out = {}
    cases = [("more", [500, 700, 300], 1024), ("tile", [100, 60, 45], 1024), ("exact", [400, 624], 1024), ("n2048", [900, 700, 600, 500], 2048),
             ("tile2048", [300, 200], 2048)]
    for ci, (tag, sizes, num_points) in enumerate(cases):
        rng = np.random.RandomState(50 + ci)
        K = len(sizes)
        parts = synthetic_parts(rng, sizes)
        norm_factor = float(rng.uniform(0.4, 0.9))
        n_total = sum(sizes)

        def fake_parser(self, f, n_max_parts, num_points, **kw):
            return (parts["nocs_p"], parts["nocs_g"], [None] * K, parts["parts_cls"], parts["parts_pts"], parts["offset_heatmap"],
                    parts["offset_unitvec"], parts["joint_orient"], parts["joint_cls"], np.zeros((K, 7), np.float32), n_total)

        ds.create_data_shape2motion = types.MethodType(fake_parser, ds)
        seed = 900 + ci
        np.random.seed(seed)
        ref = ds.create_unit_data_from_hdf5({"rgb": np.zeros((2, 2, 3), np.uint8)}, K, num_points, parts_map=[[j] for j in range(K)],
                                            norm_factors=[norm_factor], nocs_type="A")
        n_tiled = n_total if n_total >= num_points else (int(num_points / n_total) + 1) * n_total
        perm = np.random.RandomState(seed).permutation(n_tiled)
        orc = input_oracle.create_unit_data(parts, num_points, norm_factor, K, perm=perm)
        for k in orc:
            assert np.array_equal(np.asarray(ref[k]), orc[k]) and np.asarray(ref[k]).dtype == orc[k].dtype, (tag, k)
        out[f"{tag}_sizes"] = np.asarray(sizes)
        out[f"{tag}_num_points"] = np.asarray(num_points)
        out[f"{tag}_norm_factor"] = np.asarray(norm_factor)
        out[f"{tag}_perm"] = perm.astype(np.int32)
        for k, v in parts.items():
            out[f"{tag}_in_{k}"] = np.concatenate(v, 0)
        for k in orc:
            out[f"{tag}_out_{k}"] = np.asarray(ref[k])
    out["cases"] = np.asarray([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "input_sampling.npz"), **out)
    print("input_sampling.npz", os.path.getsize(os.path.join(HERE, "input_sampling.npz")))


if __name__ == "__main__":
    main()
