"""GPU: ancsh_joint_params + pose/joint_params.py against tests/golden/joint_params.npz -- values produced by the reference file's
own lines (evaluation/eval_joint_params.py:143-256, see tests/golden/gen_joint_params_golden.py).  Medians (joint points / axes) are
exact selections: equal to the reference's float32 values.  std / mean reductions run in float64 on the device against numpy's
float32 pairwise sums: bar 1e-6; camera-space joints and the two errors: bar 1e-6 (relative to values of order 1)."""
import numpy as np
import pytest
import torch

from test_joint_params_cpu import G, cases, load

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.mark.parametrize("tag", cases())
def test_joint_params_match_reference_lines(dev, tag):
    from articulated_pose_amd.pose.joint_params import joint_errors, joint_params_batch, joint_params_gt_batch
    with np.load(G) as z:
        c = load(z, tag)
    K = c["mask_pred"].shape[1]
    # a batch of two: the golden sample and the same sample with rows reversed (medians / means / std are order-free up to rounding)
    two = lambda a: np.stack([a, a[::-1].copy()])
    pred = {"gocs_per_point": two(c["gocs"]), "nocs_per_point": two(c["nocs"]), "instance_per_point": two(c["mask_pred"]),
            "heatmap_per_point": two(c["heatmap_pred"]), "unitvec_per_point": two(c["unitvec_pred"]),
            "joint_axis_per_point": two(c["orient_pred"]), "index_per_point": two(c["index_per_point"])}
    rep = lambda a: np.stack([a, a])
    out = joint_params_batch(pred, K, rep(c["pose_s"][0]), rep(c["pose_R"][0]), rep(c["pose_t"][0]), device=dev)
    o = {k: v.cpu().numpy() for k, v in out.items()}
    np.testing.assert_array_equal(o["joint_pt"][0], c["joint_p_pred"])                 # medians: the reference's float32 values
    np.testing.assert_array_equal(o["joint_axis"][0], c["joint_l_pred"])
    np.testing.assert_array_equal(o["joint_pt"][1], c["joint_p_pred"])                 # ... whatever the row order
    np.testing.assert_allclose(o["scale"][0], c["st_scale"], rtol=TOL, atol=0)
    np.testing.assert_allclose(o["translation"][0], c["st_translation"], rtol=0, atol=TOL)
    np.testing.assert_allclose(o["joint_pt_cam"][0], c["cam_p_pred"], rtol=0, atol=2 * TOL)
    np.testing.assert_allclose(o["joint_axis_cam"][0], c["cam_l_pred"], rtol=0, atol=TOL)
    gt = {"nocs_gt_g": rep(c["nocs_gt_g"]), "heatmap_gt": rep(c["heatmap_gt"]), "unitvec_gt": rep(c["unitvec_gt"]),
          "joint_axis_gt": rep(c["orient_gt"]), "joint_cls_gt": rep(c["joint_cls_gt"])}
    g = joint_params_gt_batch(gt, K, rep(c["gt_s"][0]), rep(c["gt_rt"][0]), device=dev)
    gg = {k: v.cpu().numpy() for k, v in g.items()}
    np.testing.assert_array_equal(gg["joint_pt"][0], c["joint_p_gt"])
    np.testing.assert_array_equal(gg["joint_axis"][0], c["joint_l_gt"])               # float32 row-by-row mean: numpy's order
    np.testing.assert_allclose(gg["joint_pt_cam"][0], c["cam_p_gt"], rtol=0, atol=2 * TOL)
    np.testing.assert_allclose(gg["joint_axis_cam"][0], c["cam_l_gt"], rtol=0, atol=TOL)
    ang, dist = joint_errors(out, g)
    np.testing.assert_allclose(ang.cpu().numpy()[0], c["angle_err"], rtol=0, atol=1e-4)   # degrees
    np.testing.assert_allclose(dist.cpu().numpy()[0], c["dist_err"], rtol=0, atol=1e-5)


def test_joint_params_edge_cases(dev):
    """A joint class without points gives NaN rows (np.median of an empty selection), an empty part NaN similarity; bad shapes raise."""
    from articulated_pose_amd import _lib
    from articulated_pose_amd.pose.joint_params import joint_params_batch
    rng = np.random.RandomState(0)
    K, N = 3, 100
    idx = np.zeros((1, N, K), np.float32)
    idx[..., 1] = 1.0                                   # every point votes for joint 1: joint 2 is empty
    mask = np.zeros((1, N, K), np.float32)
    mask[..., 0] = 1.0                                  # every point in part 0: parts 1, 2 empty
    pred = {"gocs_per_point": rng.rand(1, N, 3 * K).astype(np.float32), "nocs_per_point": rng.rand(1, N, 3 * K).astype(np.float32),
            "instance_per_point": mask, "heatmap_per_point": rng.rand(1, N).astype(np.float32),
            "unitvec_per_point": rng.randn(1, N, 3).astype(np.float32), "joint_axis_per_point": rng.randn(1, N, 3).astype(np.float32),
            "index_per_point": idx}
    out = joint_params_batch(pred, K, np.ones(1), np.eye(3)[None], np.zeros((1, 3)), device=dev)
    assert torch.isfinite(out["joint_pt"][0, 0]).all() and torch.isnan(out["joint_pt"][0, 1]).all() and torch.isnan(out["joint_axis"][0, 1]).all()
    assert torch.isfinite(out["scale"][0, 0]) and torch.isnan(out["scale"][0, 1:]).all() and torch.isnan(out["translation"][0, 1:]).all()
    g = torch.zeros((1, N, 5), device=dev)
    with pytest.raises(ValueError):
        _lib.call("ancsh_joint_params", 1, N, K, 5, 0, _lib.ptr(g), None, None, _lib.ptr(g), _lib.ptr(g), _lib.ptr(g), _lib.ptr(g), None, _lib.ptr(g))


def test_joint_params_single_part_object(dev):
    """K == 1: no joints -- `joint` is an empty (B, 0, 6) tensor with a null data pointer; the launcher must still return the
    similarity block of the one part instead of refusing the null pointer."""
    from articulated_pose_amd.pose.joint_params import joint_params_batch
    rng = np.random.RandomState(1)
    N = 200
    nocs = rng.rand(2, N, 3).astype(np.float32)
    pred = {"gocs_per_point": (0.7 * nocs + 0.1).astype(np.float32), "nocs_per_point": nocs, "instance_per_point": np.ones((2, N, 1), np.float32),
            "heatmap_per_point": rng.rand(2, N).astype(np.float32), "unitvec_per_point": rng.randn(2, N, 3).astype(np.float32),
            "joint_axis_per_point": rng.randn(2, N, 3).astype(np.float32), "index_per_point": np.ones((2, N, 1), np.float32)}
    out = joint_params_batch(pred, 1, np.ones(2), np.stack([np.eye(3)] * 2), np.zeros((2, 3)), device=dev)
    assert out["scale"].shape == (2, 1) and out["joint_pt"].shape == (2, 0, 3) and out["joint_pt_cam"].shape == (2, 0, 3)
    # gocs = 0.7 nocs + 0.1, and the similarity maps GLOBAL NOCS -> part NOCS (eval_joint_params.py:160-171): nocs = gocs / 0.7 - 1 / 7
    np.testing.assert_allclose(out["scale"].cpu().numpy(), 1.0 / 0.7, atol=1e-5)
    np.testing.assert_allclose(out["translation"].cpu().numpy(), -1.0 / 7.0, atol=1e-5)
