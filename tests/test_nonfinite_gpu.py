"""Non-finite input coordinates (depth sensors produce them) have DEFINED behaviour, the same on the HIP path, in the CPU oracle and
-- for the operators the reference ships as CUDA -- in the reference's own kernels compiled into oracle/_ref:

  farthest_point_sample  tf_sampling_g.cu:143-149: d2 = min(d, td) is fminf (a NaN distance leaves td alone), so a NaN / Inf point keeps
                         its initial 1e38 and is PICKED as soon as the scan reaches it -- and then picked again for every later sample
                         (no distance updates from a NaN centre).  Restated as is: indices bit-equal.
  query_ball_point       tf_grouping_g.cu:24-25: max(sqrtf(NaN), 1e-20f) = 1e-20 < radius: a NaN distance is INSIDE the ball (the point
                         of a NaN query ball is every point).  +-Inf coordinates: finite - Inf = Inf is outside, Inf - Inf = NaN inside.
  group_point / gather   copies: NaN travels with its row.
  three_nn               tf_interpolate.cpp:60-127: `d < best` is false for NaN: a NaN point is never a neighbour.
  shared-MLP layers      TensorFlow's relu / reduce_max arithmetic is third-party (absent); this build's statement is NaN-PROPAGATING
                         (nmax in csrc/common.h, orc_conv1x1 / orc_group_max in the oracle): a poisoned neighbourhood poisons every
                         output that depends on it.
  pose fit               a cloud with a non-finite value in P / part-NOCS / mask / joint axes gets an all-NaN record
                         (ancsh_pose_poison_records); the other clouds of the batch are untouched, bit for bit.
Never a silently finite answer from a poisoned cloud."""
import ctypes
import os

import numpy as np
import pytest
import torch

from helpers import cloud

pytestmark = pytest.mark.gpu
REF_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libancsh_ref_gfx950.so")


@pytest.fixture(scope="module")
def ops(dev):
    from articulated_pose_amd import tf_ops
    return tf_ops


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built")
    return ctypes.CDLL(REF_SO)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def poison(x, rng, frac, values):
    """A copy of (b, n, 3) clouds with `frac` of the points of every cloud carrying one of `values` in one random coordinate."""
    x = x.copy()
    b, n, _ = x.shape
    k = max(1, int(round(frac * n)))
    for i in range(b):
        rows = rng.choice(n, size=k, replace=False)
        x[i, rows, rng.randint(0, 3, size=k)] = rng.choice(values, size=k)
    return x


VALUES = {"nan": [np.nan], "inf": [np.inf, -np.inf], "mixed": [np.nan, np.inf, -np.inf]}


@pytest.mark.parametrize("what", ["nan", "inf", "mixed"])
@pytest.mark.parametrize("n,m", [(1024, 512), (512, 128), (2048, 512), (700, 33)])
def test_fps_with_non_finite_points(ops, oracle, ref, dev, what, n, m):
    rng = np.random.RandomState(n + m + len(what))
    x = poison(cloud(rng, 6, n, "grid"), rng, 0.01, VALUES[what])
    x[5] = cloud(rng, 1, n, "grid")[0]                                     # one clean cloud in the batch
    x[4, 0, 1] = VALUES[what][0]                                           # ... and one whose FIRST point (always sample 0) is poisoned
    want = oracle.farthest_point_sample(m, x)
    xt = T(x, dev)
    got = ops.farthest_point_sample(m, xt).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    out = torch.zeros((6, m), dtype=torch.int32, device=dev)
    temp = torch.zeros((32, n), dtype=torch.float32, device=dev)
    assert ref.ref_farthest_point_sample(6, n, m, ctypes.c_void_p(xt.data_ptr()), ctypes.c_void_p(temp.data_ptr()),
                                         ctypes.c_void_p(out.data_ptr())) == 0
    np.testing.assert_array_equal(got, out.cpu().numpy())                  # the reference's own kernel says the same
    bad = ~np.isfinite(x).all(axis=2)
    for i in range(4):                                                     # the documented degeneration: one bad point, picked again and again
        assert bad[i, got[i, 1]] and (got[i, 1:] == got[i, 1]).all()


@pytest.mark.parametrize("what", ["nan", "inf", "mixed"])
@pytest.mark.parametrize("n,m,r,ns", [(1024, 512, 0.2, 64), (512, 128, 0.4, 64), (2048, 512, 0.2, 64), (333, 77, 0.3, 16), (5000, 33, 0.15, 32)])
def test_ball_query_and_grouping_with_non_finite_points(ops, oracle, ref, dev, what, n, m, r, ns):
    rng = np.random.RandomState(3 * n + m + len(what))
    x = poison(cloud(rng, 4, n, "grid"), rng, 0.01, VALUES[what])
    q = oracle.gather_point(x, oracle.farthest_point_sample(m, x))         # the reference's own centres: mostly the poisoned point
    q[:, : m // 2] = x[:, : m // 2] if m // 2 <= n else q[:, : m // 2]     # ... half of them replaced by ordinary points of the cloud
    widx, wcnt = oracle.query_ball_point(r, ns, x, q)
    xt, qt = T(x, dev), T(q, dev)
    gidx, gcnt = ops.query_ball_point(r, ns, xt, qt)
    np.testing.assert_array_equal(gcnt.cpu().numpy(), wcnt)
    np.testing.assert_array_equal(gidx.cpu().numpy(), widx)
    idx = torch.zeros((4, m, ns), dtype=torch.int32, device=dev)
    cnt = torch.zeros((4, m), dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p
    assert ref.ref_query_ball_point(4, n, m, ctypes.c_float(r), ns, vp(xt.data_ptr()), vp(qt.data_ptr()), vp(idx.data_ptr()), vp(cnt.data_ptr())) == 0
    np.testing.assert_array_equal(gcnt.cpu().numpy(), cnt.cpu().numpy())
    np.testing.assert_array_equal(gidx.cpu().numpy(), idx.cpu().numpy())
    if what == "nan":
        nanrow = np.isnan(x).any(axis=2)
        first = np.array([np.flatnonzero(nanrow[i])[0] for i in range(4)])
        for i in range(4):                                                 # a NaN point is inside EVERY ball that still has room when the scan reaches it
            room = wcnt[i] < ns
            assert all(first[i] in widx[i, j, : wcnt[i, j]] for j in np.flatnonzero(room))
    # the multi-problem launch and the fused ball query + xyz grouping take the same decisions
    g = ops.group_point(xt, gidx).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(g), np.isnan(oracle.group_point(x, widx)))
    np.testing.assert_array_equal(np.nan_to_num(g, nan=7.0, posinf=8.0, neginf=9.0),
                                  np.nan_to_num(oracle.group_point(x, widx), nan=7.0, posinf=8.0, neginf=9.0))


@pytest.mark.parametrize("what", ["nan", "mixed"])
def test_ball_query_schedules_agree_on_non_finite_points(ops, dev, what):
    """Every ball-query kernel (wave-per-two-queries default, multi-problem launch, fused xyz grouping) decides a NaN distance alike."""
    rng = np.random.RandomState(5)
    x = poison(cloud(rng, 3, 1024, "uniform"), rng, 0.02, VALUES[what])
    q = x[:, ::2].copy()
    xt, qt = T(x, dev), T(q, dev)
    idx0, cnt0 = ops.query_ball_point(0.2, 64, xt, qt)
    assert int(cnt0.min()) >= 1                                             # every query is a point of the cloud: no untouched idx slots
    from articulated_pose_amd.tf_ops import tf_grouping
    (idx1, cnt1), = tf_grouping.query_ball_point_multi([(0.2, 64, xt, qt)])
    assert torch.equal(idx0, idx1) and torch.equal(cnt0, cnt1)
    idx2, cnt2, g2 = tf_grouping.query_ball_group_xyz(0.2, 64, xt, qt)
    assert torch.equal(idx2, idx0) and torch.equal(cnt2, cnt0)
    g0 = ops.group_point(xt, idx0)
    assert np.array_equal(g2.cpu().numpy(), g0.cpu().numpy(), equal_nan=True)
    (idx3, cnt3, g3), = tf_grouping.query_ball_group_xyz_multi([(0.2, 64, xt, qt)])
    assert torch.equal(idx3, idx0) and torch.equal(cnt3, cnt0) and np.array_equal(g3.cpu().numpy(), g0.cpu().numpy(), equal_nan=True)


@pytest.mark.parametrize("n,m", [(512, 128), (1024, 512), (77, 2)])
def test_three_nn_with_non_finite_points(ops, oracle, dev, n, m):
    rng = np.random.RandomState(n + 7 * m)
    x1 = poison(cloud(rng, 2, n, "grid"), rng, 0.02, VALUES["mixed"])
    x2 = poison(cloud(rng, 2, m, "grid"), rng, 0.05, VALUES["mixed"])
    wd, wi = oracle.three_nn(x1, x2)
    gd, gi = ops.three_nn(T(x1, dev), T(x2, dev))
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(gd.cpu().numpy(), wd)
    if oracle.have_ref_interp():
        rd, ri = oracle.ref_three_nn(x1, x2)
        np.testing.assert_array_equal(gi.cpu().numpy(), ri)
        np.testing.assert_array_equal(gd.cpu().numpy(), rd)


def _synth(rng, b, n):
    c = rng.uniform(-0.3, 0.3, (b, 1, 3))
    return (c + rng.uniform(-0.45, 0.45, (b, n, 3)) * rng.uniform(0.3, 1.0, (b, 1, 3))).astype(np.float32)


@pytest.mark.parametrize("K,N,nocs_type", [(3, 1024, "ancsh"), (3, 1024, "npcs"), (2, 2048, "ancsh")])
def test_forward_with_one_percent_nan_points(dev, K, N, nocs_type):
    """Whole forwards, 4 clouds, clouds 1 and 3 with 1 % NaN points: the HIP outputs have the ORACLE's NaN pattern (every value of a
    poisoned cloud: its global feature is NaN) and agree with it elsewhere (labels exact, floats 1e-4); the clean clouds equal, bit for
    bit, the same forward without any poisoned cloud in the batch; layer-API, grouped and paired entry points alike."""
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    mixed = nocs_type == "ancsh"
    w = synthetic_weights(K, mixed_pred=mixed, early_split_nocs=mixed, seed=K)
    rng = np.random.RandomState(K * 10 + N)
    clean = _synth(rng, 4, N)
    P = clean.copy()
    P[[1, 3]] = poison(clean[[1, 3]], rng, 0.01, [np.nan])
    want = net_oracle.forward(w, P, K, mixed_pred=mixed, early_split_nocs=mixed)
    net = Network(K, w, nocs_type, dev)
    got = {k: v.cpu().numpy() for k, v in net.predict(P).items()}
    ref_clean = {k: v.cpu().numpy() for k, v in net.predict(clean).items()}
    for k in want:
        assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), k
        assert np.isnan(got[k][[1, 3]]).all(), k                            # never a finite value out of a poisoned cloud
        assert np.isfinite(got[k][[0, 2]]).all(), k
        np.testing.assert_array_equal(got[k][[0, 2]], ref_clean[k][[0, 2]])  # the other clouds of the batch do not notice
        assert np.abs(got[k][[0, 2]] - want[k][[0, 2]]).max() <= 1e-4, k
    np.testing.assert_array_equal(got["W"][[0, 2]].argmax(2), want["W"][[0, 2]].argmax(2))
    grouped = net.predict_grouped(torch.from_numpy(P).to(dev))
    for k in got:
        assert np.array_equal(grouped[k].cpu().numpy(), got[k], equal_nan=True), k
    if mixed:
        w_n = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=20)
        pair = PairedNetworks([net, Network(K, w_n, "npcs", dev)])
        pa, _pn = pair.predict(P)
        for k in got:
            assert np.array_equal(pa[k].cpu().numpy(), got[k], equal_nan=True), (k, pair.eligible())


def test_inf_points_poison_a_cloud_too(dev):
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    K, N = 3, 1024
    w = synthetic_weights(K, seed=1)
    rng = np.random.RandomState(9)
    P = _synth(rng, 2, N)
    P[1] = poison(P[1:2], rng, 0.01, [np.inf, -np.inf])[0]
    want = net_oracle.forward(w, P, K)
    got = {k: v.cpu().numpy() for k, v in Network(K, w, "ancsh", dev).predict(P).items()}
    for k in want:
        assert np.array_equal(np.isfinite(got[k]), np.isfinite(want[k])), k
        assert not np.isfinite(got[k][1]).any() or np.array_equal(np.isnan(got[k][1]), np.isnan(want[k][1])), k
        assert np.abs(got[k][0] - want[k][0]).max() <= 1e-4, k


@pytest.mark.parametrize("K", [2, 3])
def test_pose_fit_gives_nan_records_for_poisoned_clouds(dev, K):
    """The fit of a batch in which clouds 1 (a NaN camera point), 2 (an Inf part-NOCS value) and 4 (a NaN mask row) are poisoned:
    their (K, 26) records are NaN, every other cloud's record equals the clean batch's, bit for bit; the joint-axis field poisons too."""
    from articulated_pose_amd.pose import PoseSolver
    from articulated_pose_amd.synthetic import make_cloud, make_predictions
    N, B = 512, 6
    clouds = [make_cloud(300 + i, N=N, K=K) for i in range(B)]
    preds = [make_predictions(c, K, seed=i) for i, c in enumerate(clouds)]
    st = lambda key, src: np.stack([x[key] for x in src]).copy()
    P, nocs, W, axis, cls = st("P", clouds), st("nocs_per_point", preds), st("instance_per_point", preds), st("joint_axis_per_point", preds), st("joint_cls_gt", preds)
    solver = PoseSolver(K, 0.1, 200, 16, dev)
    clean = solver.solve(P, nocs, W, axis, cls, seed=3)["record"].cpu().numpy()
    assert np.isfinite(clean).all()
    P2, nocs2, W2 = P.copy(), nocs.copy(), W.copy()
    P2[1, 17, 2] = np.nan
    nocs2[2, 400, 1] = np.inf
    W2[4, 100, :] = np.nan
    got = solver.solve(P2, nocs2, W2, axis, cls, seed=3)["record"].cpu().numpy()
    assert np.isnan(got[[1, 2, 4]]).all()
    np.testing.assert_array_equal(got[[0, 3, 5]], clean[[0, 3, 5]])
    axis2 = axis.copy()
    axis2[5, 3, 0] = -np.inf
    got = solver.solve(P, nocs, W, axis2, cls, seed=3)["record"].cpu().numpy()
    assert np.isnan(got[5]).all()
    np.testing.assert_array_equal(got[:5], clean[:5])
    a = solver.solve_stage_a(P2, nocs2, W2, seed=3)["record"].cpu().numpy()             # stage A alone: the nonlinear half was NaN anyway
    assert np.isnan(a[[1, 2, 4]]).all() and np.isfinite(a[[0, 3, 5], :, :13]).all()
    assert np.isnan(a[:, :, 13:]).all()                                                   # stage B did not run: "not fitted", never stale memory


def test_pipeline_record_of_a_poisoned_cloud_is_nan(dev):
    """End to end (both networks + fit, coupled data flow): one cloud of the batch has 1 % NaN points -> its record is NaN, the others' are finite."""
    from articulated_pose_amd.pipeline import AncshPipeline
    from articulated_pose_amd.synthetic import passthrough_pose_problem
    K, B, N = 3, 4, 1024
    pb = passthrough_pose_problem(K, B, N, seed=5)
    P = np.array(pb["P"], np.float32).copy()
    rng = np.random.RandomState(1)
    P[2] = poison(P[2:3], rng, 0.01, [np.nan])[0]
    pipe = AncshPipeline(K, pb["w_ancsh"], pb["w_npcs"], B, N, dev, couple=True, use_graph=False, niter_a=500, niter_b=16, slots=1)
    pipe.load_inputs(P, pb["cls"])
    sl, out = pipe.step()
    sl.stream.synchronize()
    rec = out["record"].cpu().numpy()
    assert np.isnan(rec[2]).all() and np.isfinite(rec[[0, 1, 3]]).all()
