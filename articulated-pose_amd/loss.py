"""Test-time losses of lib/network.py::predict_and_save (:257-316), evaluated on the MI355X.

The reference builds ~60 TensorFlow ops (lib/loss.py:54-182 via lib/network.py::compute_loss :430-498) and reduces them
per batch inside sess.run; here one kernel launch (ancsh_test_losses, csrc/loss.hip) reads every prediction / ground-truth
row once and returns the per-cloud loss tensors; collect_losses (:117-171) and the running means of predict_and_save are
scalar host arithmetic on those few numbers.  Only the inference-time configuration exists: MULTI_HEAD NOCS loss,
SELF_SU off, coord_regress_loss 'L2' (cfg/network_config.yml:66) or 'L1'."""
import ctypes

import numpy as np
import torch

from . import _lib

# cfg/network_config.yml:13-22 (the reference reads them through lib/network_config.py)
LOSS_MULTIPLIERS = dict(miou=1.0, nocs=10.0, gocs=1.0, offset=5.0, orient=0.2, index=1.0, total=1.0)
_TYPE_L = {"L2": 0, "L1": 1}


def _dev(x, dev, dtype):
    t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(dev, dtype).contiguous()


def compute_loss(pred_dict, gt_dict, n_max_parts, is_mixed, coord_regress_loss="L2"):
    """-> loss_dict of per-cloud device tensors like lib/network.py:486-496: nocs_loss (B,), miou_loss (B,K), heatmap_loss,
    unitvec_loss, orient_loss (B,), index_loss (B,3) and, when is_mixed, gocs_loss (B,).
    gt_dict keys are the batch keys fed by fill_gt_dict_with_batch_data (:373-390): nocs_gt, cls_gt, mask_array, heatmap_gt,
    unitvec_gt, orient_gt, joint_cls_gt, joint_cls_mask (+ nocs_gt_g when is_mixed)."""
    if coord_regress_loss not in _TYPE_L:
        raise ValueError("coord_regress_loss must be 'L2' or 'L1' on the test path (got %r)" % (coord_regress_loss,))
    W = pred_dict["W"]
    _lib.require_cuda(W)
    dev = W.device
    B, N, K = W.shape
    if K != n_max_parts:
        raise ValueError("W has %d part channels, n_max_parts is %d" % (K, n_max_parts))
    f32, i32 = torch.float32, torch.int32
    t = [
        _dev(W, dev, f32), _dev(pred_dict["nocs_per_point"], dev, f32),
        _dev(pred_dict["gocs_per_point"], dev, f32) if is_mixed else None,
        _dev(pred_dict["heatmap_per_point"], dev, f32), _dev(pred_dict["unitvec_per_point"], dev, f32),
        _dev(pred_dict["joint_axis_per_point"], dev, f32), _dev(pred_dict["index_per_point"], dev, f32),
        _dev(gt_dict["cls_gt"], dev, i32), _dev(gt_dict["joint_cls_gt"], dev, i32),
        _dev(gt_dict["nocs_gt"], dev, f32), _dev(gt_dict["nocs_gt_g"], dev, f32) if is_mixed else None,
        _dev(gt_dict["mask_array"], dev, f32), _dev(gt_dict["heatmap_gt"], dev, f32), _dev(gt_dict["unitvec_gt"], dev, f32),
        _dev(gt_dict["orient_gt"], dev, f32), _dev(gt_dict["joint_cls_mask"], dev, f32),
    ]
    shapes = [(B, N, K), (B, N, 3 * K), (B, N, 3 * K), (B, N, 1), (B, N, 3), (B, N, 3), (B, N, 3), (B, N), (B, N), (B, N, 3), (B, N, 3),
              (B, N, K), (B, N), (B, N, 3), (B, N, 3), (B, N)]
    for x, s in zip(t, shapes):
        if x is not None and x.numel() != int(np.prod(s)):
            raise ValueError("loss input of shape %s where %s is expected" % (tuple(x.shape), s))
    out = torch.empty((B, 5 + K + 3), dtype=f32, device=dev)
    ptrs = (ctypes.c_void_p * 16)(*[_lib.ptr(x) for x in t])
    _lib.call("ancsh_test_losses", B, N, K, _TYPE_L[coord_regress_loss], ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
    ld = {"nocs_loss": out[:, 0], "heatmap_loss": out[:, 2], "unitvec_loss": out[:, 3], "orient_loss": out[:, 4],
          "miou_loss": out[:, 5:5 + K], "index_loss": out[:, 5 + K:]}
    if is_mixed:
        ld["gocs_loss"] = out[:, 1]
    ld["_keep"] = t
    return ld


def collect_losses(loss_dict, is_mixed, pred_joint=True, pred_joint_ind=True, multipliers=LOSS_MULTIPLIERS):
    """Batch means and the weighted total, lib/network.py:117-171 -> dict of python floats keyed like predict_and_save's
    loss_dict (total_loss, total_miou_loss, total_nocs_loss, total_heatmap_loss, ...)."""
    host = {k: v.double().mean().item() for k, v in loss_dict.items() if k != "_keep"}
    t = {"total_" + k: v for k, v in host.items()}
    m = multipliers
    total = m["nocs"] * t["total_nocs_loss"] + m["miou"] * t["total_miou_loss"]
    if is_mixed:
        total += m["gocs"] * t["total_gocs_loss"]
    if pred_joint:
        if is_mixed:
            total += m["offset"] * (t["total_heatmap_loss"] + t["total_unitvec_loss"])
        total += m["orient"] * t["total_orient_loss"]
        if pred_joint_ind:
            total += m["index"] * t["total_index_loss"]
    t["total_loss"] = total * m["total"]
    return t


def reported_keys(is_mixed, pred_joint=True, early_split=True, pred_joint_ind=True):
    """The keys predict_and_save accumulates (lib/network.py:258-273), in test_loss.txt order."""
    return [key for _label, key in _fields(is_mixed, pred_joint, early_split, pred_joint_ind)]


def format_loss_result(losses, is_mixed, pred_joint=True, early_split=True, pred_joint_ind=True):
    """The line predict_and_save writes to test_loss.txt (lib/network.py:228-243), same fields in the same order."""
    return ", ".join("{}: {:6f}".format(label, losses[key]) for label, key in _fields(is_mixed, pred_joint, early_split, pred_joint_ind))


def _fields(is_mixed, pred_joint, early_split, pred_joint_ind):
    fields = [("Total Loss", "total_loss"), ("MIoU Loss", "total_miou_loss"), ("nocs Loss", "total_nocs_loss")]
    if is_mixed:
        fields.append(("gocs Loss", "total_gocs_loss"))
    if pred_joint:
        fields += [("heatmap Loss", "total_heatmap_loss"), ("unitvec Loss", "total_unitvec_loss")]
        if early_split:
            fields.append(("orient Loss", "total_orient_loss"))
    if pred_joint_ind:
        fields.append(("index Loss", "total_index_loss"))
    return fields
