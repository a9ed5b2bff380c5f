#!/usr/bin/env python3.9
"""Golden HDF5 prediction record written by the REFERENCE's own lib/prediction_io.py::save_batch_nn (:65-95).

Run with an interpreter that has h5py (this image: /opt/conda/bin/python3.9; the project interpreter has none):
    /opt/conda/bin/python3.9 tests/golden/gen_h5_golden.py
Writes tests/golden/ref_record/<basename>.h5 (two tiny clouds, N = 24 points, K = 3) plus ref_record_inputs.npz with the
arrays that went in.  The files are DATA produced by the reference code; tests/test_h5_interop_cpu.py reads them with this
build's prediction_io.load_record and re-writes them with its save_batch_nn (same interpreter, subprocess)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = ["/root/reference/lib"]


def make_batch():
    rng = np.random.RandomState(5)
    B, N, K = 2, 24, 3
    pred = {"W": rng.rand(B, N, K).astype(np.float32), "confi_per_point": rng.rand(B, N, 1).astype(np.float32),
            "nocs_per_point": rng.rand(B, N, 3 * K).astype(np.float32), "gocs_per_point": rng.rand(B, N, 3 * K).astype(np.float32),
            "heatmap_per_point": rng.rand(B, N, 1).astype(np.float32), "unitvec_per_point": rng.rand(B, N, 3).astype(np.float32),
            "joint_axis_per_point": rng.rand(B, N, 3).astype(np.float32), "index_per_point": rng.rand(B, N, 3).astype(np.float32)}
    batch = {"P": rng.rand(B, N, 3).astype(np.float32), "cls_gt": rng.randint(0, K, (B, N)).astype(np.float32),
             "nocs_gt": rng.rand(B, N, 3).astype(np.float32), "nocs_gt_g": rng.rand(B, N, 3).astype(np.float32),
             "heatmap_gt": rng.rand(B, N).astype(np.float32), "unitvec_gt": rng.rand(B, N, 3).astype(np.float32),
             "orient_gt": rng.rand(B, N, 3).astype(np.float32), "joint_cls_gt": rng.randint(0, K, (B, N)).astype(np.float32)}
    return pred, batch, ["0007_0_0", "0016_3_10"]


if __name__ == "__main__":
    import prediction_io as ref_io              # the reference module
    pred, batch, names = make_batch()
    out = os.path.join(HERE, "ref_record")
    os.makedirs(out, exist_ok=True)
    ref_io.save_batch_nn("SPFN", pred, batch, names, out, is_mixed=True, W_reduced=False)
    np.savez_compressed(os.path.join(HERE, "ref_record_inputs.npz"), names=np.asarray(names),
                        **{"pred_" + k: v for k, v in pred.items()}, **{"batch_" + k: v for k, v in batch.items()})
    for n in names:
        print(n, os.path.getsize(os.path.join(out, n + ".h5")))
