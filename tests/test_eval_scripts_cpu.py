"""CPU: oracle/eval_oracle.py against tests/golden/eval_scripts.pkl -- the printed reports and final variables of the reference's own
evaluation/eval_pose_err.py and evaluation/compute_miou.py, RUN by tests/golden/gen_eval_scripts_golden.py on a synthetic results tree.
The oracle must reproduce the variables exactly (same numpy calls in the same order) and the report tables character by character."""
import os
import pickle

import numpy as np
import pytest

from oracle import eval_oracle as EO

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["eval_scripts.pkl", "eval_scripts_drawer.pkl"])
def G(request):
    """eyeglasses (K = 3, revolute) and drawer (K = 4, prismatic: boxes posed by the URDF joint frames, instance 45841 left out of the error
    tables, the relative TRANSLATION table instead of the rotation one)"""
    with open(os.path.join(HERE, "golden", request.param), "rb") as f:
        return pickle.load(f)


def boxes_of(G):
    D, drawer = G["inputs"], G["item"] == "drawer"
    return EO.gt_boxes(D["factors"], D["corners"], sorted(D["factors"]), G["info"]["num_parts"], D["urdf"] if drawer else None,
                       G["info"]["spec_map"] if drawer else None)


def datas_of(G):
    D = G["inputs"]
    return {"pn_gt": D["gt_pn"], "gn_gt": D["gt_gn"], "baseline": D["base"], "nonlinear": D["ours"]}


def loader_of(G):
    D, info = G["inputs"], G["info"]
    return lambda exp, basename: (D["records"] if exp == info["exp"] else D["records_base"])[basename]


def tables(text):
    """the report lines of a script's stdout: table titles and rows, in order"""
    return [l for l in text.split("\n") if l.startswith(("For ", "baseline ", "nonlinea "))]


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def test_eval_pose_err_oracle_equals_the_reference_run(G):
    ref, info, K = G["eval_pose_err.py"], G["info"], G["info"]["num_parts"]
    datas, load = datas_of(G), loader_of(G)
    r_raw, t_raw = EO.raw_errors(datas, skip_instances=("45841",))
    for k in EO.KEYS:
        assert same(r_raw[k], ref["vars"]["r_raw_err"][k]) or same(r_raw[k], np.asarray(ref["vars"]["r_raw_err"][k]))
        want_t = np.asarray(ref["vars"]["t_raw_err"][k], np.float64)             # the script zeroes NaNs in place before printing
        got_t = np.asarray(t_raw[k], np.float64)
        got_t[np.isnan(got_t)] = 0
        assert same(got_t, want_t), k
    bbox = boxes_of(G)
    for ins, boxes in ref["vars"]["bbox3d_all"].items():
        assert same(bbox[ins], boxes)
    bnd = EO.boundaries(datas, load, info["exp"], info["baseline"], bbox, K)
    for k in EO.KEYS:
        assert sorted(bnd[k]) == sorted(ref["vars"]["boundary_all"][k]), k
        for name, v in ref["vars"]["boundary_all"][k].items():
            assert same(bnd[k][name]["canon"], v["canon"]) and same(bnd[k][name]["dynam"], v["dynam"]), (k, name)
    r_diff, t_diff = EO.relative_errors(datas, bnd, K)
    for k in EO.KEYS:
        assert same(r_diff[k], ref["vars"]["r_diff_raw_err"][k]) and same(t_diff[k], ref["vars"]["t_diff_raw_err"][k]), k
    lines = EO.error_report(r_raw, t_raw, K, G["domain"], "ANCSH") + EO.relative_report(r_diff, t_diff, K, G["item"], G["domain"], "ANCSH")
    assert [l for l in lines if l != "\n"] == tables(ref["stdout"])


def test_compute_miou_oracle_equals_the_reference_run(G):
    ref, info, K = G["compute_miou.py"], G["info"], G["info"]["num_parts"]
    datas, load = datas_of(G), loader_of(G)
    bbox = boxes_of(G)
    for ins, boxes in ref["vars"]["bbox3d_all"].items():
        assert same(bbox[ins], boxes)
    iou_rat, bnd = EO.miou(datas, load, info["baseline"], bbox, K)
    for k in EO.KEYS:
        assert same(iou_rat[k], ref["vars"]["iou_rat"][k]), k
        assert sorted(bnd[k]) == sorted(ref["vars"]["boundary_all"][k])
        for name, v in ref["vars"]["boundary_all"][k].items():
            assert same(bnd[k][name]["canon"], v["canon"]) and same(bnd[k][name]["dynam"], v["dynam"]), (k, name)
    assert [l for l in EO.miou_report(iou_rat, K, G["domain"], "ANCSH") if l != "\n"] == tables(ref["stdout"])
    # the fixture exercises the scripts' skip rules: a failed fit (scale None) leaves every table, a NaN translation leaves the IoU rows
    n = len(G["inputs"]["names"])
    assert len(iou_rat["baseline"]) == n - 1 and len(iou_rat["nonlinear"]) == n - 2


def test_eval_joint_params_oracle_equals_the_reference_run(G):
    ref, info, K = G["eval_joint_params.py"], G["info"], G["info"]["num_parts"]
    datas, load = datas_of(G), loader_of(G)
    angle, dist = EO.joint_param_errors(datas, load, info["exp"], K)
    assert same(np.array(angle).reshape(-1, K - 1), np.array(ref["vars"]["angle_err_all"]).reshape(-1, K - 1))
    assert same(np.array(dist).reshape(-1, K - 1), np.array(ref["vars"]["dist_err_all"]).reshape(-1, K - 1))
    assert len(angle) == len(G["inputs"]["names"]) - 1          # the failed fit (scale None) raises inside the script's try and is dropped
    tail = [l for l in ref["stdout"].split("\n") if l.strip()][-(1 + 2 * (K - 1)):]
    assert EO.joint_param_report(angle, dist, K) == tail
