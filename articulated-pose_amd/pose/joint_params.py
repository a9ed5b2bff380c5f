"""Joint parameters from the networks' per-point heads, batched on the GPU: the per-sample body of
evaluation/eval_joint_params.py (:143-256) -- similarity global NOCS -> part NOCS per part, offset voting
`nocs_g + unitvec * (1 - heatmap) * 0.2` with per-joint medians, transformation into camera space by part 0's fitted pose,
and the two joint errors -- for a whole batch in one kernel launch (ancsh_joint_params, csrc/metrics.hip) plus a handful of
(B, K-1, 3) float64 tensor operations.  The metric functions are pose/metrics.py's (lib/d3_utils.py:137-142,165-174)."""
import torch

from .. import _lib
from .metrics import axis_diff_degree_batch, dist_between_3d_lines_batch


def _f32(t, dev):
    return torch.as_tensor(t, dtype=torch.float32, device=dev).contiguous() if not torch.is_tensor(t) else t.to(dev, torch.float32).contiguous()


def _launch(gocs, nocs, mask, heatmap, unitvec, axis, joint_cls, K, axis_mean):
    dev = gocs.device
    _lib.require_cuda(gocs)
    B, N, G = gocs.shape
    st = torch.empty((B, K, 4), dtype=torch.float64, device=dev) if nocs is not None else None
    joint = torch.empty((B, max(K - 1, 0), 6), dtype=torch.float64, device=dev)
    if K > 1 or nocs is not None:
        _lib.call("ancsh_joint_params", B, N, K, G, 1 if axis_mean else 0, _lib.ptr(gocs), _lib.ptr(nocs), _lib.ptr(mask), _lib.ptr(heatmap),
                  _lib.ptr(unitvec), _lib.ptr(axis), _lib.ptr(joint_cls), _lib.ptr(st), _lib.ptr(joint))
    return st, joint


def joint_params_batch(pred, num_parts, pose_scale, pose_rotation, pose_translation, device="cuda:0"):
    """pred: the record's prediction arrays (B, N, .) -- gocs_per_point (3K or 3 channels), nocs_per_point (3K),
    instance_per_point (K), heatmap_per_point (1 or none), unitvec_per_point (3), joint_axis_per_point (3), index_per_point (K);
    pose_*: part 0's fitted pose per cloud (the 'nonlinear' entries of the pose pickle, :201-203): scale (B,), rotation (B,3,3),
    translation (B,3).  Returns float64 device tensors:
      scale (B,K), translation (B,K,3)            st_dict of :160-171
      joint_pt, joint_axis (B,K-1,3)              joints['pred'] in global-NOCS space (:176-187)
      joint_pt_cam, joint_axis_cam (B,K-1,3)      t_joints['pred'] in camera space (:214-222)"""
    dev = torch.device(device)
    K = num_parts
    g = _f32(pred["gocs_per_point"], dev)
    B, N = g.shape[:2]
    jc = torch.argmax(_f32(pred["index_per_point"], dev), dim=2).to(torch.int32).contiguous()       # np.argmax(index_per_point, 1) (:134)
    st, joint = _launch(g, _f32(pred["nocs_per_point"], dev), _f32(pred["instance_per_point"], dev),
                        _f32(pred["heatmap_per_point"], dev).reshape(B, N), _f32(pred["unitvec_per_point"], dev),
                        _f32(pred["joint_axis_per_point"], dev), jc, K, False)
    out = {"scale": st[..., 0], "translation": st[..., 1:], "joint_pt": joint[..., :3], "joint_axis": joint[..., 3:]}
    s2, t2 = st[:, 0, 0], st[:, 0, 1:]                                           # part 0 is the platform (:215-217)
    R = torch.as_tensor(pose_rotation, dtype=torch.float64, device=dev).reshape(B, 3, 3)
    s = torch.as_tensor(pose_scale, dtype=torch.float64, device=dev).reshape(B, 1, 1)
    t = torch.as_tensor(pose_translation, dtype=torch.float64, device=dev).reshape(B, 1, 3)
    p_part = out["joint_pt"] * s2[:, None, None] + t2[:, None, :]
    out["joint_pt_cam"] = (s * p_part) @ R.transpose(1, 2) + t                   # np.dot(s[0] * p, r[0].T) + t[0]
    out["joint_axis_cam"] = out["joint_axis"] @ R.transpose(1, 2)
    return out


def joint_params_gt_batch(gt, num_parts, gt_scale, gt_rt, device="cuda:0"):
    """gt: nocs_gt_g (B,N,3), heatmap_gt (B,N), unitvec_gt, joint_axis_gt (B,N,3), joint_cls_gt (B,N); gt_scale (B,), gt_rt (B,4,4):
    part 0's ground-truth global-NOCS pose (:205-206).  -> joint_pt / joint_axis (global NOCS; the axis is the MEAN, :195) and
    their camera-space images (:224-231)."""
    dev = torch.device(device)
    g = _f32(gt["nocs_gt_g"], dev)
    B, N = g.shape[:2]
    jc = torch.as_tensor(gt["joint_cls_gt"], device=dev).to(torch.int32).reshape(B, N).contiguous()
    _st, joint = _launch(g, None, None, _f32(gt["heatmap_gt"], dev).reshape(B, N), _f32(gt["unitvec_gt"], dev),
                         _f32(gt["joint_axis_gt"], dev), jc, num_parts, True)
    rt = torch.as_tensor(gt_rt, dtype=torch.float64, device=dev).reshape(B, 4, 4)
    s = torch.as_tensor(gt_scale, dtype=torch.float64, device=dev).reshape(B, 1, 1)
    R = rt[:, :3, :3]
    return {"joint_pt": joint[..., :3], "joint_axis": joint[..., 3:],
            "joint_pt_cam": (s * joint[..., :3]) @ R.transpose(1, 2) + rt[:, None, :3, 3],
            "joint_axis_cam": joint[..., 3:] @ R.transpose(1, 2)}


def joint_errors(pred, gt):
    """(angle_err, dist_err), each (B, K-1): axis_diff_degree and dist_between_3d_lines of the camera-space joints (:244-256)."""
    return (axis_diff_degree_batch(gt["joint_axis_cam"], pred["joint_axis_cam"]),
            dist_between_3d_lines_batch(gt["joint_pt_cam"], gt["joint_axis_cam"], pred["joint_pt_cam"], pred["joint_axis_cam"]))
