"""CPU: oracle/joint_params_oracle.py against tests/golden/joint_params.npz (the reference's own lines, see the generator):
every intermediate the reference keeps (st_dict, joints, t_joints, the two errors), exactly."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "joint_params.npz")


def cases():
    with np.load(G) as z:
        return [str(c) for c in z["cases"]]


def load(z, tag):
    return {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_") and not (tag == "k3" and k.startswith("k3_shared_gn_"))}


@pytest.mark.parametrize("tag", cases())
def test_joint_params_oracle_equals_reference_lines(tag):
    from oracle import joint_params_oracle as JO
    from oracle import metrics_oracle as MO
    with np.load(G) as z:
        c = load(z, tag)
    K = c["mask_pred"].shape[1]
    jc = np.argmax(c["index_per_point"], axis=1)
    sc, tr, p, l = JO.st_and_joints(c["gocs"], c["nocs"], c["mask_pred"], c["heatmap_pred"], c["unitvec_pred"], c["orient_pred"], jc, K)
    np.testing.assert_array_equal(sc.astype(np.float64), c["st_scale"])
    np.testing.assert_array_equal(tr.astype(np.float64), c["st_translation"])
    np.testing.assert_array_equal(p.astype(np.float64), c["joint_p_pred"])
    np.testing.assert_array_equal(l.astype(np.float64), c["joint_l_pred"])
    pg, lg = JO.gt_joints(c["nocs_gt_g"], c["heatmap_gt"], c["unitvec_gt"], c["orient_gt"], c["joint_cls_gt"], K)
    np.testing.assert_array_equal(pg.astype(np.float64), c["joint_p_gt"])
    np.testing.assert_array_equal(lg.astype(np.float64), c["joint_l_gt"])
    cp, cl = JO.to_camera_pred(p, l, sc[0], tr[0], c["pose_s"][0], c["pose_R"][0], c["pose_t"][0])
    np.testing.assert_array_equal(cp, c["cam_p_pred"])
    np.testing.assert_array_equal(cl, c["cam_l_pred"])
    gp, gl = JO.to_camera_gt(pg, lg, c["gt_s"][0], c["gt_rt"][0])
    np.testing.assert_array_equal(gp, c["cam_p_gt"])
    np.testing.assert_array_equal(gl, c["cam_l_gt"])
    for j in range(K - 1):
        assert MO.axis_diff_degree(gl[j], cl[j]) == c["angle_err"][j]
        assert MO.dist_between_3d_lines(gp[j], gl[j], cp[j], cl[j]) == c["dist_err"][j]
