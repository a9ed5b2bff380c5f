"""GPU test of the drop-in entry points: main --test writes prediction records with the reference's key
names, pose_multi_process fits them and writes the reference's pickle schema."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_main_test_then_pose_multi_process(dev, tmp_path, monkeypatch):
    from articulated_pose_amd import main as main_mod, pose_multi_process, prediction_io
    from articulated_pose_amd.synthetic import make_cloud, make_predictions
    base = tmp_path
    data = base / "data"
    data.mkdir()
    names = []
    for inst, art, frame in (("0007", "0", "0"), ("0007", "0", "5"), ("0016", "3", "10")):
        c = make_cloud(len(names), N=1024, K=3)
        name = f"{inst}_{art}_{frame}"
        names.append(name)
        np.savez(data / (name + ".npz"), P=c["P"], cls_gt=c["cls_gt"], nocs_gt=c["nocs_gt"], joint_cls_gt=c["cls_gt"])
    monkeypatch.setenv("ANCSH_BASE_PATH", str(base))
    for nocs_type, exp in (("ancsh", "3.9"), ("npcs", "3.91")):
        main_mod.main(["--item", "eyeglasses", "--nocs_type", nocs_type, "--test", "--data_dir", str(data),
                       "--out_dir", str(base / "results/test_pred" / exp), "--batch_size", "2"])
    rec = prediction_io.load_record(str(base / "results/test_pred/3.9"), names[0])
    for k in ("P", "nocs_per_point", "instance_per_point", "gocs_per_point", "confidence_per_point", "heatmap_per_point",
              "unitvec_per_point", "joint_axis_per_point", "index_per_point", "joint_cls_gt", "cls_gt", "nocs_gt"):
        assert k in rec, k
    assert rec["instance_per_point"].shape == (1024, 3) and rec["nocs_per_point"].shape == (1024, 9)
    # overwrite the baseline records' predictions with usable ones (random-init heads are degenerate)
    for i, n in enumerate(names):
        c = make_cloud(i, N=1024, K=3)
        p = make_predictions(c, 3, seed=i)
        r = prediction_io.load_record(str(base / "results/test_pred/3.91"), n)
        r.update(nocs_per_point=p["nocs_per_point"], instance_per_point=p["instance_per_point"])
        np.savez(base / "results/test_pred/3.91" / (n + ".npz"), **r)
        ra = prediction_io.load_record(str(base / "results/test_pred/3.9"), n)
        ra.update(joint_axis_per_point=p["joint_axis_per_point"])
        np.savez(base / "results/test_pred/3.9" / (n + ".npz"), **ra)
    pose_multi_process.main(["--item", "eyeglasses", "--domain", "unseen", "--nocs", "ANCSH", "--base_path", str(base)])
    sub = base / "results/pickle/3.9/subs/3.91_unseen_ANCSH_eyeglasses_rt_ours_0.1_0.pkl"
    out = pickle.load(open(sub, "rb"))
    assert set(out) == set(names)
    r = out[names[0]]
    assert set(r) == {"scale", "rotation", "translation", "xyz_err", "rpy_err", "scale_err"}
    assert len(r["rotation"]["baseline"]) == 3 and len(r["rotation"]["nonlinear"]) == 3
    c0 = make_cloud(0, N=1024, K=3)
    from articulated_pose_amd.pose.d3_utils import rot_diff_degree
    for j in range(3):
        assert rot_diff_degree(r["rotation"]["nonlinear"][j], c0["R"][j]) < 3.0
