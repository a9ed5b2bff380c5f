"""CPU: oracle/loss_oracle.py (float32 restatement of lib/loss.py's test-time path; TensorFlow code: wiring pinned by tests/test_loss_trace_cpu.py, op arithmetic from the TF definitions)
cross-checked against an independent float64 evaluation of the formulas in lib/loss.py:54-182."""
import numpy as np
import pytest

from oracle import loss_oracle as LO


def fake_batch(B, N, K, seed, mixed=True):
    rng = np.random.RandomState(seed)
    sm = lambda x: np.exp(x) / np.exp(x).sum(-1, keepdims=True)
    cls = rng.randint(-1 if seed % 2 else 0, K, (B, N))
    pred = dict(W=sm(rng.randn(B, N, K)).astype(np.float32), nocs_per_point=rng.rand(B, N, 3 * K).astype(np.float32),
                heatmap_per_point=rng.rand(B, N, 1).astype(np.float32), unitvec_per_point=np.tanh(rng.randn(B, N, 3)).astype(np.float32),
                joint_axis_per_point=np.tanh(rng.randn(B, N, 3)).astype(np.float32), index_per_point=sm(rng.randn(B, N, 3)).astype(np.float32))
    mask = np.zeros((B, N, K), np.float32)
    np.put_along_axis(mask, np.where(cls < 0, K - 1, cls)[..., None], 1.0, axis=2)
    jcls = rng.randint(0, 3, (B, N))
    gt = dict(cls_gt=cls, nocs_gt=rng.rand(B, N, 3).astype(np.float32), mask_array=mask, heatmap_gt=rng.rand(B, N).astype(np.float32),
              unitvec_gt=rng.randn(B, N, 3).astype(np.float32), orient_gt=rng.randn(B, N, 3).astype(np.float32), joint_cls_gt=jcls,
              joint_cls_mask=(jcls > 0).astype(np.float32))
    if mixed:
        pred["gocs_per_point"] = rng.rand(B, N, 3 * K).astype(np.float32)
        gt["nocs_gt_g"] = rng.rand(B, N, 3).astype(np.float32)
    return pred, gt


@pytest.mark.parametrize("K,mixed,type_l", [(3, True, "L2"), (2, False, "L2"), (4, True, "L1")])
def test_loss_oracle_vs_float64(K, mixed, type_l):
    pred, gt = fake_batch(3, 257, K, seed=K, mixed=mixed)
    ld = LO.loss_dict(pred, gt, K, mixed, type_l)
    d = lambda a, b: np.linalg.norm(a - b, axis=-1) if type_l == "L2" else np.abs(a - b).sum(-1)
    P64 = {k: np.asarray(v, np.float64) for k, v in pred.items()}
    G64 = {k: np.asarray(v, np.float64) for k, v in gt.items()}
    want_nocs = sum((G64["mask_array"][:, :, i] * d(P64["nocs_per_point"][:, :, 3 * i:3 * i + 3], G64["nocs_gt"])).mean(1) for i in range(K))
    np.testing.assert_allclose(ld["nocs_loss"], want_nocs, rtol=2e-6)
    np.testing.assert_allclose(ld["heatmap_loss"], (np.abs(P64["heatmap_per_point"][..., 0] - G64["heatmap_gt"]) * G64["joint_cls_mask"]).mean(1), rtol=2e-6)
    np.testing.assert_allclose(ld["orient_loss"], (d(P64["joint_axis_per_point"], G64["orient_gt"]) * G64["joint_cls_mask"]).mean(1), rtol=2e-6)
    onehot = (gt["cls_gt"][..., None] == np.arange(K)).astype(np.float64)
    dot = (onehot * P64["W"]).sum(1)
    np.testing.assert_allclose(ld["miou_loss"], 1 - dot / (onehot.sum(1) + P64["W"].sum(1) - dot + 1e-10), rtol=1e-5, atol=1e-6)
    tot = LO.collect_losses(ld, mixed)
    assert ("total_gocs_loss" in tot) == mixed and np.isfinite(tot["total_loss"])
    expect = 10 * ld["nocs_loss"].mean() + ld["miou_loss"].mean() + 0.2 * ld["orient_loss"].mean() + ld["index_loss"].mean()
    if mixed:
        expect += ld["gocs_loss"].mean() + 5 * (ld["heatmap_loss"].mean() + ld["unitvec_loss"].mean())
    assert abs(tot["total_loss"] - expect) < 1e-5
