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
        # the ground-truth fields of a reference record (lib/prediction_io.py:73-92): main --test then also writes test_loss.txt
        rng = np.random.RandomState(len(names))
        np.savez(data / (name + ".npz"), P=c["P"], cls_gt=c["cls_gt"], nocs_gt=c["nocs_gt"], joint_cls_gt=c["cls_gt"],
                 nocs_gt_g=c["nocs_gt"], heatmap_gt=rng.rand(1024).astype(np.float32), unitvec_gt=rng.randn(1024, 3).astype(np.float32),
                 joint_axis_gt=np.tile(c["joint_axis"], (1024, 1)).astype(np.float32))
    monkeypatch.setenv("ANCSH_BASE_PATH", str(base))
    for nocs_type, exp in (("ancsh", "3.9"), ("npcs", "3.91")):
        main_mod.main(["--item", "eyeglasses", "--nocs_type", nocs_type, "--test", "--data_dir", str(data),
                       "--out_dir", str(base / "results/test_pred" / exp), "--batch_size", "2"])
    for exp in ("3.9", "3.91"):      # ground truth present in the records -> the loss line of lib/network.py:228-243
        txt = open(base / "results/test_pred" / exp / "test_loss.txt").read()
        assert txt.startswith("Total Loss: ") and "MIoU Loss: " in txt and ("gocs Loss" in txt) == (exp == "3.9")
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


def test_compute_gt_pose_matches_reference_golden_and_feeds_pose_multi_process(dev, tmp_path):
    """compute_gt_pose --save writes the rts_all pickle (compose_rt over Umeyama, evaluation/compute_gt_pose.py:14-19,82-97)
    equal to the reference-generated golden, and pose_multi_process then fills the 'gt' / *_err entries from it."""
    from articulated_pose_amd import compute_gt_pose, pose_multi_process, prediction_io
    from articulated_pose_amd.synthetic import make_cloud, make_predictions
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gt_pose.npz"))
    recs = [("0007_0_%d" % (5 * i), dict(P=g[f"P{i}"], nocs_gt=g[f"nocs_gt{i}"], cls_gt=g[f"cls_gt{i}"])) for i in range(int(g["n_cases"]))]
    out0 = compute_gt_pose.gt_pose_records(recs[:1], 3, "ANCSH", dev)
    out1 = compute_gt_pose.gt_pose_records(recs[1:], 4, "ANCSH", dev)
    for i, (out, K) in enumerate(((out0, 3), (out1, 4))):
        r = out[recs[i][0]]
        assert set(r) == {"scale", "rt"} and len(r["rt"]["gt"]) == K and r["rt"]["gt"][0].dtype == np.float32
        np.testing.assert_allclose(np.stack(r["rt"]["gt"]), g[f"rt{i}"], atol=1e-4)
        np.testing.assert_allclose(np.stack(r["scale"]["gt"]), g[f"scale{i}"], atol=1e-4)
    # a record with an empty part is skipped (the reference's bare except), the others still come out
    bad = dict(recs[0][1], cls_gt=np.zeros_like(recs[0][1]["cls_gt"]))
    assert list(compute_gt_pose.gt_pose_records([("bad_0_0", bad), recs[0]], 3, "ANCSH", dev)) == [recs[0][0]]
    # the CLI: record files -> pickle (protocol 2) -> pose_multi_process reads it as rts_all
    base = tmp_path
    for exp in ("3.9", "3.91"):
        (base / "results/test_pred" / exp).mkdir(parents=True)
    names = []
    for i, (inst, art, frame) in enumerate((("0007", "0", "0"), ("0016", "3", "10"), ("0001", "0", "0"))):
        c = make_cloud(20 + i, N=1024, K=3)
        p = make_predictions(c, 3, seed=i)
        name = f"{inst}_{art}_{frame}"
        names.append(name)
        np.savez(base / "results/test_pred/3.9" / (name + ".npz"), P=c["P"], cls_gt=c["cls_gt"], nocs_gt=c["nocs_gt"],
                 joint_cls_gt=c["cls_gt"], joint_axis_per_point=p["joint_axis_per_point"])
        np.savez(base / "results/test_pred/3.91" / (name + ".npz"), P=c["P"], nocs_per_point=p["nocs_per_point"],
                 instance_per_point=p["instance_per_point"])
    rts = compute_gt_pose.main(["--item", "eyeglasses", "--domain", "unseen", "--nocs", "ANCSH", "--save", "--base_path", str(base)])
    assert set(rts) == set(names[:2])                      # 0001 is a seen instance
    pk = base / "results/pickle/3.9/unseen_ANCSH_eyeglasses_rt.pkl"
    loaded = pickle.load(open(pk, "rb"))
    c0 = make_cloud(20, N=1024, K=3)
    for j in range(3):                                      # noise-free GT NOCS: Umeyama recovers the synthetic pose
        np.testing.assert_allclose(loaded[names[0]]["rt"]["gt"][j][:3, :3], c0["R"][j], atol=1e-3)
        np.testing.assert_allclose(loaded[names[0]]["scale"]["gt"][j][0], c0["s"][j], rtol=1e-3)
    pose_multi_process.main(["--item", "eyeglasses", "--domain", "unseen", "--nocs", "ANCSH", "--base_path", str(base)])
    res = pickle.load(open(base / "results/pickle/3.9/subs/3.91_unseen_ANCSH_eyeglasses_rt_ours_0.1_0.pkl", "rb"))
    r = res[names[0]]
    assert len(r["rotation"]["gt"]) == 3 and len(r["rpy_err"]["nonlinear"]) == 3 and len(r["scale_err"]["baseline"]) == 3
    assert max(r["rpy_err"]["nonlinear"]) < 3.0 and max(r["xyz_err"]["nonlinear"]) < 0.05

    # the reference's own parallel entry forks its workers itself (evaluation/pose_multi_process.py:53-67); so does this one:
    # a plain `python -m ... --gpus 2` starts two ranks (both on the one GPU of a test box), each writing its slice's pickle
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    for f in (base / "results/pickle/3.9/subs").iterdir():
        f.unlink()
    rr = subprocess.run([sys.executable, "-m", "articulated_pose_amd.pose_multi_process", "--item", "eyeglasses", "--domain", "unseen",
                         "--nocs", "ANCSH", "--base_path", str(base), "--gpus", "2"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stderr[-3000:]
    subs = sorted(f.name for f in (base / "results/pickle/3.9/subs").iterdir())
    assert subs == ["3.91_unseen_ANCSH_eyeglasses_rt_ours_0.1_0.pkl", "3.91_unseen_ANCSH_eyeglasses_rt_ours_0.1_1.pkl"]
    got = {}
    for f in subs:
        got.update(pickle.load(open(base / "results/pickle/3.9/subs" / f, "rb")))
    assert set(got) == set(names[:2])                      # the two ranks' slices cover the unseen test group


def test_evaluation_sh_end_to_end(dev, tmp_path, capsys):
    """The reference's evaluation.sh on this build, step by step on synthetic records of four frames: compute_gt_pose (both NOCS
    types) -> pose_multi_process -> eval_pose_err -> compute_miou -> eval_joint_params.  Two things evaluation.sh expects to find
    already are synthesised: the baseline pickle *_rt_pn.pkl (written by the reference's baseline_npcs.py, outside evaluation.sh; here
    a copy of this path's records, which carry their own 'baseline' entries) and the dataset's factor / corner tables."""
    from articulated_pose_amd import compute_gt_pose, compute_miou, eval_joint_params, eval_pose_err, pose_multi_process
    from articulated_pose_amd.synthetic import make_cloud, make_predictions
    base, K, N = tmp_path, 3, 1024
    for exp in ("3.9", "3.91"):
        (base / "results/test_pred" / exp).mkdir(parents=True)
    names, clouds = [], []
    for i, (inst, art, frame) in enumerate((("0007", "0", "0"), ("0007", "2", "5"), ("0016", "3", "10"), ("0036", "1", "25"))):
        c = make_cloud(40 + i, N=N, K=K)
        p = make_predictions(c, K, seed=i)
        rng = np.random.RandomState(i)
        name = f"{inst}_{art}_{frame}"
        names.append(name)
        clouds.append(c)
        gocs = (c["nocs_gt"] * 0.8 + 0.1).astype(np.float32)
        jc = np.where(rng.rand(N) < 0.5, c["cls_gt"], 0)               # points of part j vote for joint j (part 0's points: no joint)
        common = dict(P=c["P"], cls_gt=c["cls_gt"].astype(np.float32), nocs_gt=c["nocs_gt"], nocs_gt_g=gocs, joint_cls_gt=jc.astype(np.float32),
                      heatmap_gt=rng.rand(N).astype(np.float32), unitvec_gt=rng.randn(N, 3).astype(np.float32),
                      joint_axis_gt=np.tile(c["joint_axis"], (N, 1)).astype(np.float32), instance_per_point=p["instance_per_point"].astype(np.float32),
                      nocs_per_point=p["nocs_per_point"].astype(np.float32),
                      gocs_per_point=np.tile(gocs, (1, K)) + rng.randn(N, 3 * K).astype(np.float32) * 0.003,
                      heatmap_per_point=rng.rand(N).astype(np.float32), unitvec_per_point=rng.randn(N, 3).astype(np.float32),
                      joint_axis_per_point=p["joint_axis_per_point"].astype(np.float32),
                      index_per_point=(np.eye(K)[jc] * 0.7 + rng.rand(N, K) * 0.3).astype(np.float32))
        for exp in ("3.9", "3.91"):
            np.savez(base / "results/test_pred" / exp / (name + ".npz"), **common)
    argv = ["--item", "eyeglasses", "--domain", "unseen", "--base_path", str(base)]
    for nocs in ("ANCSH", "NAOCS"):                                          # step 1 (the scripts read both ground-truth pickles)
        assert set(compute_gt_pose.main(argv + ["--nocs", nocs, "--save"])) == set(names)
    pose_multi_process.main(argv + ["--nocs", "ANCSH"])                      # step 2
    pk = base / "results/pickle/3.9"
    ours = pickle.load(open(pk / "subs/3.91_unseen_ANCSH_eyeglasses_rt_ours_0.1_0.pkl", "rb"))
    assert set(ours) == set(names)
    pickle.dump(ours, open(pk / "unseen_ANCSH_eyeglasses_rt_pn.pkl", "wb"))
    ds = base / "shape2motion/pickle"
    ds.mkdir(parents=True)
    factors = {ins: [1.0] * (K + 1) for ins in ("0007", "0016", "0036")}
    corners = {ins: [np.stack([np.full((1, 3), -0.35), np.full((1, 3), 0.35)]) for _ in range(K + 1)] for ins in ("0007", "0016", "0036")}
    pickle.dump(factors, open(ds / "eyeglasses.pkl", "wb"))
    pickle.dump(corners, open(ds / "eyeglasses_corners.pkl", "wb"))
    capsys.readouterr()
    e = eval_pose_err.main(argv + ["--nocs", "ANCSH"])                       # step 3
    text = capsys.readouterr().out
    assert "mean rotation err per part" in text and "5 degrees, 5 cms accuracy per part" in text and "mean relative rotation err per part" in text
    r = np.asarray(e["r_raw_err"]["nonlinear"])
    assert r.shape == (4, K) and r.max() < 5.0 and np.asarray(e["t_raw_err"]["nonlinear"]).max() < 0.05
    rd = np.asarray(e["r_diff_raw_err"]["nonlinear"])
    assert rd.shape == (4, K - 1) and np.isfinite(rd).all() and rd.max() < 8.0        # relative rotation of two fits that are each < 5 degrees off
    assert set(e["boundary_all"]["nonlinear"]) == set(names)
    m = compute_miou.main(argv + ["--nocs", "ANCSH"])                        # step 4
    text = capsys.readouterr().out
    assert "3D IoU per part" in text
    iou = np.asarray(m["iou_rat"]["nonlinear"])
    assert iou.shape == (4, K) and np.isfinite(iou).all() and (iou >= 0).all() and (iou <= 1).all() and iou.max() > 0.05
    j = eval_joint_params.main(argv + ["--nocs", "ANCSH"])                   # step 5
    text = capsys.readouterr().out
    assert "joint 0 with mean angle error" in text and "joint 1 with mean angle error" in text
    assert j["angle_err_all"].shape == (4, K - 1) and np.isfinite(j["angle_err_all"]).all() and j["angle_err_all"].max() < 5.0
    assert np.isfinite(j["dist_err_all"]).all()
