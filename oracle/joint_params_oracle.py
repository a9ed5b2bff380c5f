"""CPU restatement of the joint-parameter extraction of evaluation/eval_joint_params.py (TEST INFRASTRUCTURE ONLY).
Pinned against tests/golden/joint_params.npz, whose values come from the reference file's own lines 143-256 executed as they lie
(tests/golden/gen_joint_params_golden.py).  float32 inputs; numpy evaluates the element arithmetic in float32 like the reference."""
import numpy as np


def st_and_joints(gocs, nocs, mask_pred, heatmap, unitvec, orient, joint_cls, K):
    """One sample.  -> (scale (K,), translation (K,3), joint_pt (K-1,3), joint_axis (K-1,3)) as the reference's float32 values."""
    cls = np.argmax(mask_pred, axis=1)                                            # :151
    gn = np.zeros((gocs.shape[0], 3), np.float32)
    pn = np.zeros((gocs.shape[0], 3), np.float32)
    scale, trans = [], []
    for j in range(K):                                                            # :160-171
        idx = np.where(cls == j)[0]
        pn[idx] = nocs[idx, j * 3:j * 3 + 3]
        gn[idx] = gocs[idx, :3] if gocs.shape[1] == 3 else gocs[idx, j * 3:j * 3 + 3]
        x, y = gn[idx], pn[idx]
        s = np.std(np.mean(y, axis=1)) / np.std(np.mean(x, axis=1))
        scale.append(s)
        trans.append(np.mean(y - s * x, axis=0))
    pts, axes = [], []
    for j in range(1, K):                                                         # :176-187
        offset = unitvec * (1 - heatmap.reshape(-1, 1)) * 0.2
        joint_pts = gn + offset
        idx = np.where(joint_cls == j)[0]
        axes.append(np.median(orient[idx], axis=0))
        pts.append(np.median(joint_pts[idx], axis=0))
    return np.array(scale), np.stack(trans), np.stack(pts), np.stack(axes)


def gt_joints(nocs_gt_g, heatmap_gt, unitvec_gt, orient_gt, joint_cls_gt, K):      # :189-199
    pts, axes = [], []
    for j in range(1, K):
        offset = unitvec_gt * (1 - heatmap_gt.reshape(-1, 1)) * 0.2
        joint_pts = nocs_gt_g + offset
        idx = np.where(joint_cls_gt == j)[0]
        axes.append(np.mean(orient_gt[idx], axis=0))
        pts.append(np.median(joint_pts[idx], axis=0))
    return np.stack(pts), np.stack(axes)


def to_camera_pred(joint_pt, joint_axis, s2, t2, s0, R0, t0):                     # :214-222
    """s0 reaches the reference as a float32 scalar (scale_pts of float32 arrays) or a Python float: either way `s[0] * p` is a
    float32 product; the rotation / translation that follow are float64."""
    ps, ls = [], []
    for j in range(joint_pt.shape[0]):                   # joint by joint like the reference ((1,3) x (3,3) products)
        p = joint_pt[j] * s2 + t2
        ps.append((np.dot(np.float32(s0) * p.reshape(1, 3), R0.T) + t0).reshape(3))
        ls.append(np.dot(joint_axis[j].reshape(1, 3), R0.T).reshape(3))
    return np.stack(ps), np.stack(ls)


def to_camera_gt(joint_pt, joint_axis, s_g, rt_g):                                # :224-231
    ps = [(np.dot(np.float32(s_g) * joint_pt[j].reshape(1, 3), rt_g[:3, :3].T) + rt_g[:3, 3]).reshape(3) for j in range(joint_pt.shape[0])]
    ls = [np.dot(joint_axis[j].reshape(1, 3), rt_g[:3, :3].T).reshape(3) for j in range(joint_pt.shape[0])]
    return np.stack(ps), np.stack(ls)
