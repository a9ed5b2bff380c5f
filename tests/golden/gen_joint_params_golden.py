"""Generator of tests/golden/joint_params.npz -- BUILD CONTAINER ONLY (needs /root/reference).

The joint-parameter extraction of evaluation/eval_joint_params.py is a script body, not a function: lines 143-256 (part / joint
index lists, global-NOCS -> part-NOCS similarity per part, offset voting `nocs + unitvec * (1 - heatmap) * 0.2`, per-joint medians,
transformation into camera space by part 0's pose, axis / line-distance errors) sit inside a loop over .h5 files.  This generator
EXECUTES THOSE LINES AS THEY LIE in the reference file -- read at run time, dedented, exec'd in a namespace that holds the
variables the preceding lines (116-141) would have read from the .h5 record and the pose pickles -- so the golden values come from
the reference's own numpy statements, not from a restatement.  Nothing of the reference is written to disk; the .npz holds inputs
and results only.  A guard checks the range still starts / ends on the expected statements.

    python tests/golden/gen_joint_params_golden.py        # rewrites tests/golden/joint_params.npz
"""
import io
import os
import sys
import textwrap
import types
from contextlib import redirect_stdout

import numpy as np

REF = os.environ.get("ANCSH_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
FIRST, LAST = 143, 256


def reference_block():
    lines = open(os.path.join(REF, "evaluation", "eval_joint_params.py")).read().splitlines()
    assert lines[FIRST - 1].strip() == "part_idx_list_gt   = []", lines[FIRST - 1]
    assert lines[LAST - 1].strip() == "dist_err.append(t_diff)", lines[LAST - 1]
    return compile(textwrap.dedent("\n".join(lines[FIRST - 1:LAST])), "eval_joint_params.py:%d-%d" % (FIRST, LAST), "exec")


def d3_utils():
    """lib/d3_utils.py imports h5py / matplotlib-free helpers; only its two metric functions are needed (pure numpy)."""
    for name in ("h5py", "cv2", "trimesh", "descartes"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    try:
        from lib import d3_utils as D
    finally:
        sys.path.remove(REF)
    return D


def rot(rng):
    q, _ = np.linalg.qr(rng.randn(3, 3))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def make_case(seed, N, K, gn_per_part):
    """A synthetic .h5 record + pose pickles (float32 arrays with the reference's keys and shapes)."""
    rng = np.random.RandomState(seed)
    f = np.float32
    lab = rng.randint(0, K, N)
    mask_gt = lab.astype(f)                                                    # cls_gt
    mask_pred = (np.eye(K)[lab] * 0.7 + rng.rand(N, K) * 0.4).astype(f)        # instance_per_point (a few flips)
    nocs_gt = {"gn": rng.rand(N, 3).astype(f), "pn": rng.rand(N, 3).astype(f)}
    nocs_pred = {"pn": rng.rand(N, 3 * K).astype(f),
                 "gn": rng.rand(N, 3 * K if gn_per_part else 3).astype(f)}
    jc = rng.randint(0, K, N)
    index_per_point = (np.eye(K)[jc] * 0.6 + rng.rand(N, K) * 0.5).astype(f)
    joint_cls_gt = jc.astype(f)
    rt = lambda: np.vstack([np.hstack([rot(rng), rng.randn(3, 1)]), [[0, 0, 0, 1]]])
    base = "0001_0_0"
    datas = {"pn_gt": {base: {"rt": {"gt": [rt() for _ in range(K)]}, "scale": {"gt": [np.array([1 + rng.rand()] * 3) for _ in range(K)]}}},
             "gn_gt": {base: {"rt": {"gt": [rt() for _ in range(K)]}, "scale": {"gt": [1 + rng.rand() for _ in range(K)]}}},
             "nonlinear": {base: {"rotation": {"nonlinear": [rot(rng) for _ in range(K)]},
                                   "translation": {"nonlinear": [rng.randn(3) for _ in range(K)]},
                                   "scale": {"nonlinear": [1 + rng.rand() for _ in range(K)]}}}}
    return dict(num_parts=K, basename=base, datas=datas, mask_gt=mask_gt, mask_pred=mask_pred, nocs_gt=nocs_gt, nocs_pred=nocs_pred,
                heatmap_pred=rng.rand(N).astype(f), heatmap_gt=rng.rand(N).astype(f),
                unitvec_pred=rng.randn(N, 3).astype(f), unitvec_gt=rng.randn(N, 3).astype(f),
                orient_pred=rng.randn(N, 3).astype(f), orient_gt=rng.randn(N, 3).astype(f),
                joint_cls_pred=np.argmax(index_per_point, axis=1), index_per_point=index_per_point, joint_cls_gt=joint_cls_gt)


def main():
    block, D = reference_block(), d3_utils()
    out = {}
    cases = [("k3", 1, 512, 3, True), ("k2", 2, 333, 2, True), ("k4", 3, 1024, 4, True), ("k3_shared_gn", 4, 256, 3, False)]
    for tag, seed, N, K, gpp in cases:
        ns = make_case(seed, N, K, gpp)
        ns.update(np=np, axis_diff_degree=D.axis_diff_degree, dist_between_3d_lines=D.dist_between_3d_lines,
                  angle_err_all=[], dist_err_all=[])
        with redirect_stdout(io.StringIO()):
            exec(block, ns)
        K1 = K - 1
        pre = tag + "_"
        for k in ("mask_pred", "heatmap_pred", "heatmap_gt", "unitvec_pred", "unitvec_gt", "orient_pred", "orient_gt", "index_per_point",
                  "joint_cls_gt"):
            out[pre + k] = ns[k]
        out[pre + "gocs"], out[pre + "nocs"] = ns["nocs_pred"]["gn"], ns["nocs_pred"]["pn"]
        out[pre + "nocs_gt_g"] = ns["nocs_gt"]["gn"]
        nl = ns["datas"]["nonlinear"][ns["basename"]]
        out[pre + "pose_R"] = np.stack(nl["rotation"]["nonlinear"])
        out[pre + "pose_t"] = np.stack(nl["translation"]["nonlinear"])
        out[pre + "pose_s"] = np.array(nl["scale"]["nonlinear"])
        out[pre + "gt_rt"] = np.stack(ns["datas"]["gn_gt"][ns["basename"]]["rt"]["gt"])
        out[pre + "gt_s"] = np.array(ns["datas"]["gn_gt"][ns["basename"]]["scale"]["gt"])
        # results of the reference's statements
        out[pre + "st_scale"] = np.array(ns["st_dict"]["scale"], np.float64)
        out[pre + "st_translation"] = np.stack(ns["st_dict"]["translation"]).astype(np.float64)
        for kind in ("pred", "gt"):
            out[pre + "joint_p_" + kind] = np.stack([j["p"] for j in ns["joints"][kind]]).astype(np.float64)
            out[pre + "joint_l_" + kind] = np.stack([j["l"] for j in ns["joints"][kind]]).astype(np.float64)
            out[pre + "cam_p_" + kind] = np.stack([j["p"].reshape(3) for j in ns["t_joints"][kind]]).astype(np.float64)
            out[pre + "cam_l_" + kind] = np.stack([j["l"].reshape(3) for j in ns["t_joints"][kind]]).astype(np.float64)
        out[pre + "angle_err"] = np.array(ns["angle_err"], np.float64)
        out[pre + "dist_err"] = np.array(ns["dist_err"], np.float64)
        assert len(ns["angle_err"]) == K1
        print(tag, "joints", K1, "angle_err", ns["angle_err"], "dist_err", ns["dist_err"])
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "joint_params.npz"), **out)


if __name__ == "__main__":
    main()
