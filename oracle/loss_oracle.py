"""CPU ORACLE of the test-time losses (test infrastructure only; never imported by the product).

numpy float32 restatement, op by op, of what lib/network.py::compute_loss (:430-498, called with is_eval=False) and
collect_losses (:117-171) evaluate for test_loss.txt:
    lib/loss.py:54-102  compute_nocs_loss(MULTI_HEAD=True, SELF_SU=False)
    lib/loss.py:104-166 compute_vect_loss(confidence=joint_cls_mask, MULTI_HEAD=False, SELF_SU=False)
    lib/loss.py:169-182 compute_miou_loss(W, I_gt)  (no Hungarian reordering: network.py:465)
PARITY: the WIRING is pinned -- tests/golden/loss_trace.json is the op trace the reference's own lib/network.py (compute_loss,
collect_losses) + lib/loss.py leave under a recording tensorflow stand-in, and tests/test_loss_trace_cpu.py interprets it in numpy float32
and requires this file to equal it.  The ARITHMETIC of the TensorFlow ops is not (tensorflow-gpu==1.10.1 is absent here and the reference
holds no fixture for it): tf.norm = sqrt(sum(square)), tf.one_hot(-1) = zero row, reduce_mean over the point axis follow the TF op
definitions; cross-checked against a float64 evaluation in tests/test_loss_cpu.py."""
import numpy as np

DIVISION_EPS = np.float32(1e-10)        # lib/constants.py:1

# cfg/network_config.yml:13-22
MULTIPLIERS = dict(miou=1.0, nocs=10.0, gocs=1.0, offset=5.0, orient=0.2, index=1.0, total=1.0)


def _f(x):
    return np.asarray(x, np.float32)


def compute_nocs_loss(nocs, nocs_gt, num_parts, mask_array, TYPE_L="L2"):
    """lib/loss.py:61-73 (MULTI_HEAD, not SELF_SU).  nocs (B,N,3K), nocs_gt (B,N,3), mask_array (B,N,K) -> (B,)"""
    nocs, nocs_gt, mask_array = _f(nocs), _f(nocs_gt), _f(mask_array)
    loss = np.zeros(nocs.shape[0], np.float32)
    for i in range(num_parts):
        d = nocs[:, :, 3 * i:3 * i + 3] - nocs_gt
        diff_l2 = np.sqrt(np.sum(d * d, axis=2, dtype=np.float32))
        diff_abs = np.sum(np.abs(d), axis=2, dtype=np.float32)
        diff = diff_l2 if TYPE_L == "L2" else diff_abs
        loss = loss + np.mean(mask_array[:, :, i] * diff, axis=1, dtype=np.float32)
    return loss


def compute_vect_loss(vect, vect_gt, confidence, TYPE_L="L2"):
    """lib/loss.py:138-166 (not MULTI_HEAD, not SELF_SU).  vect (B,N,1|3), confidence (B,N) -> (B,)"""
    vect, vect_gt, confidence = _f(vect), _f(vect_gt), _f(confidence)
    if vect.shape[2] == 1:
        diff_l2 = diff_abs = np.abs(vect[:, :, 0] - vect_gt) * confidence
    else:
        d = vect - vect_gt
        diff_l2 = np.sqrt(np.sum(d * d, axis=2, dtype=np.float32)) * confidence
        diff_abs = np.sum(np.abs(d), axis=2, dtype=np.float32) * confidence
    return np.mean(diff_l2 if TYPE_L == "L2" else diff_abs, axis=1, dtype=np.float32)


def compute_miou_loss(W, I_gt):
    """lib/loss.py:169-182.  W (B,N,K), I_gt (B,N) int (-1 = unassigned) -> (B,K)"""
    W = _f(W)
    K = W.shape[2]
    W_gt = (np.asarray(I_gt)[..., None] == np.arange(K)).astype(np.float32)      # tf.one_hot: -1 -> zero row
    dot = np.sum(W_gt * W, axis=1, dtype=np.float32)
    denominator = np.sum(W_gt, axis=1, dtype=np.float32) + np.sum(W, axis=1, dtype=np.float32) - dot
    return np.float32(1.0) - dot / (denominator + DIVISION_EPS)


def loss_dict(pred, gt, num_parts, is_mixed, TYPE_L="L2"):
    """lib/network.py:462-498: per-cloud loss tensors."""
    out = {
        "miou_loss": compute_miou_loss(pred["W"], gt["cls_gt"]),
        "nocs_loss": compute_nocs_loss(pred["nocs_per_point"], gt["nocs_gt"], num_parts, gt["mask_array"], TYPE_L),
        "heatmap_loss": compute_vect_loss(pred["heatmap_per_point"], gt["heatmap_gt"], gt["joint_cls_mask"], TYPE_L),
        "unitvec_loss": compute_vect_loss(pred["unitvec_per_point"], gt["unitvec_gt"], gt["joint_cls_mask"], TYPE_L),
        "orient_loss": compute_vect_loss(pred["joint_axis_per_point"], gt["orient_gt"], gt["joint_cls_mask"], TYPE_L),
        "index_loss": compute_miou_loss(pred["index_per_point"], gt["joint_cls_gt"]),
    }
    if is_mixed:
        out["gocs_loss"] = compute_nocs_loss(pred["gocs_per_point"], gt["nocs_gt_g"], num_parts, gt["mask_array"], TYPE_L)
    return out


def collect_losses(ld, is_mixed, pred_joint=True, pred_joint_ind=True, mult=MULTIPLIERS):
    """lib/network.py:117-171: batch means and the weighted total."""
    t = {"total_nocs_loss": np.mean(ld["nocs_loss"]), "total_miou_loss": np.mean(ld["miou_loss"]),
         "total_heatmap_loss": np.mean(ld["heatmap_loss"]), "total_unitvec_loss": np.mean(ld["unitvec_loss"]),
         "total_orient_loss": np.mean(ld["orient_loss"]), "total_index_loss": np.mean(ld["index_loss"])}
    total = mult["nocs"] * t["total_nocs_loss"] + mult["miou"] * t["total_miou_loss"]
    if is_mixed:
        t["total_gocs_loss"] = np.mean(ld["gocs_loss"])
        total += mult["gocs"] * t["total_gocs_loss"]
    if pred_joint:
        if is_mixed:
            total += mult["offset"] * t["total_heatmap_loss"] + mult["offset"] * t["total_unitvec_loss"]
        total += mult["orient"] * t["total_orient_loss"]
        if pred_joint_ind:
            total += mult["index"] * t["total_index_loss"]
    t["total_loss"] = total * mult["total"]
    return {k: float(v) for k, v in t.items()}
