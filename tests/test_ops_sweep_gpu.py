"""GPU: randomised shape sweep of the PointNet++ operators against the CPU oracle -- the fixed-shape tests of test_ops_gpu.py cover the
network's shapes and a handful of odd ones; here every operator meets ~40 seeded random shapes (batch 1..5, clouds of 1..3000 points,
1..300 queries, nsample 1..96, 0..140 channels, radii from "nothing in the ball" to "everything in the ball", all four cloud kinds),
bit-exact as everywhere else.  One seed per case: a failure names the case and is reproducible."""
import numpy as np
import pytest
import torch

from helpers import cloud

pytestmark = pytest.mark.gpu

KINDS = ["uniform", "grid", "coarse", "tiled"]
# 40 seeds per operator in the suite; ANCSH_SWEEP_SEEDS=1000 for a one-off long fuzz (profiles/r05_ops_fuzz.txt); seeds repeat the
# suite's pattern modulo 40 (the last 8 of every 40 reach the large-cloud kernels)
import os
SEEDS = range(int(os.environ.get("ANCSH_SWEEP_SEEDS", "40")))


@pytest.fixture(scope="module")
def ops(dev):
    from articulated_pose_amd import tf_ops
    return tf_ops


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _size(rng, hi):
    """sizes biased towards the small and the awkward: 1, 2, 3, powers of two +- 1, and a uniform tail"""
    pick = rng.randint(4)
    if pick == 0:
        return int(rng.randint(1, min(hi, 8) + 1))
    if pick == 1:
        return int(np.clip(2 ** rng.randint(1, 12) + rng.randint(-1, 2), 1, hi))
    return int(rng.randint(1, hi + 1))


@pytest.mark.parametrize("seed", SEEDS)
def test_fps_and_gather(ops, oracle, dev, seed):
    rng = np.random.RandomState(1000 + seed)
    b, n = int(rng.randint(1, 6)), _size(rng, 20000 if seed % 40 >= 32 else 3000)        # the last seeds reach the large-cloud kernel
    m = _size(rng, min(n, 600))
    x = cloud(rng, b, n, KINDS[seed % 4])
    want = oracle.farthest_point_sample(m, x)
    got = ops.farthest_point_sample(m, T(x, dev))
    np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg="b=%d n=%d m=%d" % (b, n, m))
    np.testing.assert_array_equal(ops.gather_point(T(x, dev), got).cpu().numpy(), oracle.gather_point(x, want))


@pytest.mark.parametrize("seed", SEEDS)
def test_ball_query_and_group(ops, oracle, dev, seed):
    rng = np.random.RandomState(2000 + seed)
    b, n, m = int(rng.randint(1, 6)), _size(rng, 12000 if seed % 40 >= 32 else 3000), _size(rng, 300)   # > 5120 points: the unstaged path
    ns = _size(rng, 96)
    r = float([0.01, 0.1, 0.2, 0.4, 1.0, 5.0][rng.randint(6)])
    kind = KINDS[seed % 4]
    x = cloud(rng, b, n, kind)
    q = cloud(rng, b, m, kind)
    if seed % 3 == 0 and m <= n:
        q = x[:, rng.permutation(n)[:m]].copy()                  # queries that ARE points of the cloud (distance 0 hits)
    wi, wc = oracle.query_ball_point(r, ns, x, q)
    gi, gc = ops.query_ball_point(r, ns, T(x, dev), T(q, dev))
    msg = "b=%d n=%d m=%d ns=%d r=%g %s" % (b, n, m, ns, r, kind)
    np.testing.assert_array_equal(gc.cpu().numpy(), wc, err_msg=msg)
    np.testing.assert_array_equal(gi.cpu().numpy(), wi, err_msg=msg)
    c = int([0, 1, 3, 4, 6, 61, 64, 128, 131, 140][rng.randint(10)])
    pts = rng.randn(b, n, c).astype(np.float32)
    np.testing.assert_array_equal(ops.group_point(T(pts, dev), gi).cpu().numpy(), oracle.group_point(pts, wi), err_msg=msg + " c=%d" % c)
    # the fused forms give the same tensors
    fi, fc, fg = ops.query_ball_group_xyz(r, ns, T(x, dev), T(q, dev), center=bool(seed & 1))
    assert torch.equal(fi, gi) and torch.equal(fc, gc), msg
    want_g = oracle.group_point(x, wi) - (q[:, :, None, :] if seed & 1 else 0)
    np.testing.assert_array_equal(fg.cpu().numpy(), want_g.astype(np.float32), err_msg=msg)


@pytest.mark.parametrize("seed", SEEDS)
def test_three_nn_and_interpolate(ops, oracle, dev, seed):
    from articulated_pose_amd.tf_ops.tf_interpolate import three_weights
    rng = np.random.RandomState(3000 + seed)
    b, n, m = int(rng.randint(1, 6)), _size(rng, 3000), _size(rng, 700)
    kind = KINDS[seed % 4]
    x1, x2 = cloud(rng, b, n, kind), cloud(rng, b, m, kind)
    if seed % 3 == 0:
        k = min(n, m)
        x1[:, :k] = x2[:, :k]                                     # coincident points: zero distances, the 1e-10 clamp of the weights
    wd, wi = oracle.three_nn(x1, x2)
    gd, gi = ops.three_nn(T(x1, dev), T(x2, dev))
    msg = "b=%d n=%d m=%d %s" % (b, n, m, kind)
    np.testing.assert_array_equal(gi.cpu().numpy(), wi, err_msg=msg)
    np.testing.assert_array_equal(gd.cpu().numpy(), wd, err_msg=msg)
    c = int([1, 3, 4, 7, 64, 128, 140][rng.randint(7)])
    pts = rng.randn(b, m, c).astype(np.float32)
    w = three_weights(gd)
    got = ops.three_interpolate(T(pts, dev), gi, w).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.three_interpolate(pts, wi, w.cpu().numpy()), err_msg=msg + " c=%d" % c)


@pytest.mark.parametrize("seed", SEEDS)
def test_conv1x1(oracle, dev, seed):
    """ancsh_conv1x1 on random (rows, cin, cout, activation, row stride): every routing of the shared-MLP layer (row-tile, packed,
    few-rows, generic workgroup-tiled kernel) must give the oracle's k-ordered fmaf chain bit for bit."""
    from test_mlp_gpu import make_layer, run_gpu
    rng = np.random.RandomState(4000 + seed)
    rows = _size(rng, 5000)
    cin = int([1, 3, 5, 64, 128, 131, 256, 259, 384, 512, 1280][rng.randint(11)]) if seed % 2 else _size(rng, 300)
    cout = int([1, 3, 9, 10, 16, 64, 128, 256, 512, 1024][rng.randint(10)]) if seed % 3 else _size(rng, 300)
    act = int(rng.randint(2))
    x = rng.randn(rows, cin).astype(np.float32)
    layer = make_layer(rng, cin, cout, bn=bool(act))
    ldx = cin if seed % 4 else ((cin + 3) // 4) * 4 + 4 * int(rng.randint(3))
    got = run_gpu(x, layer, act, dev, ldx=ldx)
    np.testing.assert_array_equal(got, oracle.conv1x1(x, layer, act), err_msg="rows=%d cin=%d cout=%d act=%d ldx=%d" % (rows, cin, cout, act, ldx))
