"""GPU: out-of-bounds evidence for the whole C ABI (VERDICT r04 "what's weak" 9).  Every output and scratch buffer the host layer
allocates -- it does so with torch.empty, without exception -- is carved out of a sentinel-filled allocation (tests/redzone.py:
4 KB of 0xA5 before and after, the body pre-filled with NaN / -1) while the existing parity checks run: the operators on their seeded
random shapes, the conv / chain / fused-SA entry points on ragged row counts, the whole forward of both networks, the pose fit on
ragged and empty parts, the metric / input / loss kernels.  A write past any buffer fails the test; an element a kernel failed to
write shows up as NaN / -1 in the parity check itself.  Negative control: an output one row too small IS caught."""
import numpy as np
import pytest
import torch

from helpers import cloud
from redzone import RedzoneError, guarded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from articulated_pose_amd import tf_ops
    return tf_ops


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_negative_control_a_row_too_few_is_caught(dev):
    """ancsh_group_point told that `out` holds m rows per cloud while the buffer holds m - 1: the last row lands in the redzone."""
    from articulated_pose_amd import _lib
    rng = np.random.RandomState(0)
    b, n, c, m, ns = 2, 100, 7, 9, 16
    pts, idx = T(rng.randn(b, n, c).astype(np.float32), dev), T(rng.randint(0, n, (b, m, ns)).astype(np.int32), dev)
    with pytest.raises(RedzoneError, match="PAST the buffer"):
        with guarded():
            out = torch.empty((b * m - 1, ns, c), dtype=torch.float32, device=dev)          # one row short
            _lib.call("ancsh_group_point", b, n, c, m, ns, _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(out))
    with guarded() as arena:                                                                    # the exact size passes
        out = torch.empty((b * m, ns, c), dtype=torch.float32, device=dev)
        _lib.call("ancsh_group_point", b, n, c, m, ns, _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(out))
        assert arena.check() == 1 and not torch.isnan(out).any()
    with pytest.raises(RedzoneError, match="BEFORE the buffer"):                               # and a write in front of a buffer
        with guarded():
            out = torch.empty((4, 4), dtype=torch.float32, device=dev)
            torch.as_strided(out, (1,), (1,), storage_offset=out.storage_offset() - 1).fill_(1.0)


@pytest.mark.parametrize("seed", range(0, 40, 3))
def test_operator_sweep_under_redzones(ops, oracle, dev, seed):
    import test_ops_sweep_gpu as S
    with guarded() as arena:
        S.test_fps_and_gather(ops, oracle, dev, seed)
        S.test_ball_query_and_group(ops, oracle, dev, seed)
        S.test_three_nn_and_interpolate(ops, oracle, dev, seed)
        S.test_conv1x1(oracle, dev, seed)
        assert arena.check() >= 8


@pytest.mark.parametrize("seed", [1, 6, 11, 33])
def test_unaligned_buffers(ops, oracle, dev, seed):
    """The operator entry points promise no alignment: the same checks with every buffer 4 bytes off a 16-byte boundary
    (the float4 / dwordx3 paths must take their scalar fallbacks, not fault or write a rounded-down address)."""
    import test_ops_sweep_gpu as S
    with guarded(misalign=4):
        S.test_fps_and_gather(ops, oracle, dev, seed)
        rng = np.random.RandomState(2000 + seed)
        b, n, m, ns = 3, 1000 + seed, 77, 24
        x, q = cloud(rng, b, n, "uniform"), cloud(rng, b, m, "uniform")
        gi, gc = ops.query_ball_point(0.3, ns, T(x, dev), T(q, dev))
        wi, wc = oracle.query_ball_point(0.3, ns, x, q)
        np.testing.assert_array_equal(gi.cpu().numpy(), wi)
        np.testing.assert_array_equal(gc.cpu().numpy(), wc)
        for c in (3, 7, 64):
            pts = rng.randn(b, n, c).astype(np.float32)
            np.testing.assert_array_equal(ops.group_point(T(pts, dev), gi).cpu().numpy(), oracle.group_point(pts, wi))
        S.test_three_nn_and_interpolate(ops, oracle, dev, seed)


@pytest.mark.parametrize("K,N,B", [(3, 777, 5), (2, 1000, 3), (4, 2048, 2), (3, 1024, 8)])
def test_network_forward_under_redzones(dev, K, N, B):
    """Both forwards (layer API and the grouped / chained launches) with every activation, index and scratch buffer guarded; outputs
    equal to the unguarded run bit for bit (so no NaN body pattern survives in any head tensor)."""
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from test_network_gpu import synth_cloud
    P = torch.from_numpy(synth_cloud(np.random.RandomState(K * 10 + B), B, N)).to(dev)
    a = Network(K, synthetic_weights(K, seed=0), "ancsh", dev)
    n = Network(K, synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1), "npcs", dev)
    plain = [a.predict(P), n.predict(P)]
    with guarded() as arena:
        alone = [a.predict(P), n.predict(P)]
        pair = PairedNetworks([a, n]).predict(P)
        assert arena.check() > 40
    for g in range(2):
        for k in plain[g]:
            assert torch.equal(alone[g][k], plain[g][k]) and torch.equal(pair[g][k], plain[g][k]), (g, k)


def test_chains_on_ragged_rows_under_redzones(dev):
    """the tail chains (one- and two-tile), the mid-section chains and the grouped conv launches on row counts that are not multiples
    of their 32-row tiles, unaligned inputs included"""
    import test_mid_chain_gpu as MC
    import test_mlp_gpu as M
    import test_network_gpu as NW
    with guarded() as arena:
        for rows, cin, ldx, offset in [(1, 131, 132, 0), (33, 131, 131, 0), (257, 131, 132, 1)]:
            M.test_mlp_chain_program_equals_layer_by_layer(dev, rows, cin, ldx, offset)
        for rows, ldx in [(1000, 132), (33, 131)]:
            M.test_mlp_chain_grouped_one_tile_equals_layer_by_layer(dev, rows, ldx)
        for G, B in [(1, 1), (2, 3), (3, 5)]:
            MC.test_mid_chains_equal_layer_by_layer(dev, G, B)
        MC.test_each_chain_against_its_layers(dev)
        NW.test_grouped_conv_equals_plain_bitwise(dev)
        assert arena.check() > 30


@pytest.mark.parametrize("seed", [0, 3, 5, 6, 9, 12, 17, 21])
def test_pose_fit_under_redzones(dev, seed):
    """PoseSolver.solve on the sweep's ragged problems (parts of two dozen points, K = 2 / 3 / 4, odd budgets) and with a part nobody
    is predicted as (empty part -> NaN rows, best = -1): every model, mask, score, record and scratch array guarded; results equal to
    the unguarded solve."""
    from articulated_pose_amd.pose import PoseSolver
    from articulated_pose_amd.pose.parallel_ancsh_pose import draws_from_seed
    from test_pose_sweep_gpu import _problem
    c, p, K, na, nb = _problem(seed)
    W = p["instance_per_point"].copy()
    if seed % 4 == 1:
        W[:, K - 1] = -1.0                                  # nobody in the last part
    counts = np.bincount(np.argmax(W, 1), minlength=K)
    da, db = draws_from_seed(500 + seed, np.maximum(counts, 1), na, nb)
    args = (c["P"][None], p["nocs_per_point"][None], W[None], p["joint_axis_per_point"][None], p["joint_cls_gt"][None], da[None], db[None])
    keys = ("record", "best_a", "best_b", "score_b", "inliers_a", "inliers_b", "tie_a", "tie_b", "counts", "labels")
    plain = PoseSolver(K, 0.1, na, nb, dev).solve(*args)
    plain = {k: plain[k].cpu().numpy() for k in keys}
    with guarded() as arena:
        sol = PoseSolver(K, 0.1, na, nb, dev).solve(*args)
        assert arena.check() >= 20
    for k in keys:
        np.testing.assert_array_equal(sol[k].cpu().numpy(), plain[k], err_msg=k)
    if seed % 4 == 1:
        assert np.isnan(plain["record"][0, K - 1]).all() and plain["best_a"][0, K - 1, 0] == -1


def test_metric_input_and_loss_kernels_under_redzones(dev, oracle):
    import test_eval_scripts_gpu as E
    import test_input_gpu as I
    import test_joint_params_gpu as J
    import test_loss_gpu as L
    import test_metrics_gpu as Mx
    with guarded() as arena:
        Mx.test_iou_3d_random_vs_oracle_and_amodal_boxes(dev)
        Mx.test_scalar_metrics_golden(dev)
        for tag in J.cases():
            J.test_joint_params_match_reference_lines(dev, tag)
        J.test_joint_params_edge_cases(dev)
        I.test_input_sample_batch_equals_reference_records(dev)
        I.test_input_sample_device_rng_invariants(dev, oracle)
        for cfg in [(3, 257, 3, True, "L2"), (5, 64, 4, True, "L1")]:
            L.test_losses_match_oracle(dev, *cfg)
        for cfg in [(3, 257, 3, 9), (1, 2048, 8, 24)]:
            E.test_part_extents_kernel(dev, *cfg)
        assert arena.check() > 10
