"""GPU parity of the PointNet++ operators: HIP (through the C ABI) vs the CPU oracle, bit-exact
for indices / counts / copies, and vs the reference's own .cu kernels compiled into oracle/_ref."""
import ctypes
import os

import numpy as np
import pytest
import torch

from helpers import cloud

pytestmark = pytest.mark.gpu

REF_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                      "libancsh_ref_gfx950.so")


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


@pytest.mark.parametrize("kind", ["uniform", "grid", "coarse", "tiled"])
@pytest.mark.parametrize("n,m", [(1024, 512), (512, 128), (2048, 512), (700, 33), (64, 64), (5, 3)])
def test_fps_matches_oracle(ops, oracle, dev, kind, n, m):
    rng = np.random.RandomState(n * 7 + m)
    x = cloud(rng, 3, n, kind)
    want = oracle.farthest_point_sample(m, x)
    got = ops.farthest_point_sample(m, T(x, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("kind", ["uniform", "coarse"])
def test_fps_large_cloud_kernel(ops, oracle, ref, dev, kind):
    """n > 8192: the large-cloud kernel (running minima in the caller's scratch) against the oracle and the reference's own
    kernel; without the scratch the C ABI refuses with EINVAL instead of computing something else."""
    from articulated_pose_amd import _lib
    from articulated_pose_amd.tf_ops.tf_sampling import farthest_point_sample_gather
    rng = np.random.RandomState(77)
    n, m = 9000, 40
    x = cloud(rng, 2, n, kind)
    xt = T(x, dev)
    want = oracle.farthest_point_sample(m, x)
    np.testing.assert_array_equal(ops.farthest_point_sample(m, xt).cpu().numpy(), want)
    idx, xyz = farthest_point_sample_gather(m, xt)
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    np.testing.assert_array_equal(xyz.cpu().numpy(), oracle.gather_point(x, want))
    out = torch.zeros((2, m), dtype=torch.int32, device=dev)
    temp = torch.zeros((32, n), dtype=torch.float32, device=dev)
    assert ref.ref_farthest_point_sample(2, n, m, ctypes.c_void_p(xt.data_ptr()), ctypes.c_void_p(temp.data_ptr()),
                                         ctypes.c_void_p(out.data_ptr())) == 0
    if kind == "coarse":                       # lattice input: exact arithmetic, the reference kernel is directly comparable
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    with pytest.raises(ValueError):
        _lib.call("ancsh_farthest_point_sample", 2, n, m, _lib.ptr(xt), 0, _lib.ptr(out))


@pytest.mark.parametrize("n,m", [(10, 5), (1000, 64), (8192, 100), (9001, 33), (20000, 257)])
def test_prob_sample(ops, oracle, ref, dev, n, m):
    """ProbSample: indices AND the cumulative sums (order-dependent float additions) equal to the oracle's simulation of the
    reference block scan and to the reference's own kernels (oracle/_ref), bit for bit."""
    from articulated_pose_amd import _lib
    rng = np.random.RandomState(n + m)
    p = rng.rand(3, n).astype(np.float32)
    r = rng.rand(3, m).astype(np.float32)
    want, want_cdf = oracle.prob_sample(p, r)
    pt, rt = T(p, dev), T(r, dev)
    np.testing.assert_array_equal(ops.prob_sample(pt, rt).cpu().numpy(), want)
    temp = torch.zeros((3, n), dtype=torch.float32, device=dev)
    out = torch.zeros((3, m), dtype=torch.int32, device=dev)
    _lib.call("ancsh_prob_sample", 3, n, m, _lib.ptr(pt), _lib.ptr(rt), _lib.ptr(temp), _lib.ptr(out))
    np.testing.assert_array_equal(temp.cpu().numpy(), want_cdf)
    rtemp, rout = torch.zeros_like(temp), torch.zeros_like(out)
    vp = ctypes.c_void_p
    assert ref.ref_prob_sample(3, n, m, vp(pt.data_ptr()), vp(rt.data_ptr()), vp(rtemp.data_ptr()), vp(rout.data_ptr())) == 0
    np.testing.assert_array_equal(rtemp.cpu().numpy(), want_cdf)
    np.testing.assert_array_equal(rout.cpu().numpy(), want)
    with pytest.raises(ValueError):
        ops.prob_sample(pt[0], rt)


def test_fps_gather_fused(ops, oracle, dev):
    from articulated_pose_amd.tf_ops.tf_sampling import farthest_point_sample_gather
    rng = np.random.RandomState(1)
    x = cloud(rng, 4, 1024)
    idx, xyz = farthest_point_sample_gather(512, T(x, dev))
    want = oracle.farthest_point_sample(512, x)
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    np.testing.assert_array_equal(xyz.cpu().numpy(), oracle.gather_point(x, want))
    np.testing.assert_array_equal(ops.gather_point(T(x, dev), idx).cpu().numpy(), oracle.gather_point(x, want))


@pytest.mark.parametrize("kind", ["uniform", "grid", "coarse", "tiled"])
@pytest.mark.parametrize("n,m,r,ns", [(1024, 512, 0.2, 64), (512, 128, 0.4, 64), (2048, 512, 0.2, 64),
                                      (333, 77, 0.3, 16), (100, 10, 5.0, 64), (100, 10, 0.01, 8),
                                      (4096, 50, 0.15, 64), (5000, 33, 0.15, 32)])
def test_ball_query_matches_oracle(ops, oracle, dev, kind, n, m, r, ns):
    rng = np.random.RandomState(n + m)
    x = cloud(rng, 3, n, kind)
    q = oracle.gather_point(x, oracle.farthest_point_sample(m, x)) if m <= n else cloud(rng, 3, m, kind)
    widx, wcnt = oracle.query_ball_point(r, ns, x, q)
    gidx, gcnt = ops.query_ball_point(r, ns, T(x, dev), T(q, dev))
    np.testing.assert_array_equal(gcnt.cpu().numpy(), wcnt)
    np.testing.assert_array_equal(gidx.cpu().numpy(), widx)


def test_ball_query_multi_equals_separate_calls(ops, oracle, dev):
    """Both SA levels' ball queries (plus two odd-shaped problems) in ONE launch: every output equal to the separate ops."""
    rng = np.random.RandomState(3)
    x = cloud(rng, 5, 1024, "coarse")
    l1 = oracle.gather_point(x, oracle.farthest_point_sample(512, x))
    l2 = oracle.gather_point(l1, oracle.farthest_point_sample(128, l1))
    y = cloud(rng, 2, 333, "uniform")
    probs = [(0.2, 64, T(x, dev), T(l1, dev)), (0.4, 64, T(l1, dev), T(l2, dev)), (0.3, 16, T(y, dev), T(y[:, :77].copy(), dev)),
             (5.0, 8, T(y, dev), T(y[:, :1].copy(), dev))]
    got = ops.query_ball_point_multi(probs)
    for (r, ns, a, q), (gi, gc) in zip(probs, got):
        wi, wc = ops.query_ball_point(r, ns, a, q)
        assert torch.equal(gi, wi) and torch.equal(gc, wc)
        oi, oc = oracle.query_ball_point(r, ns, a.cpu().numpy(), q.cpu().numpy())
        np.testing.assert_array_equal(gi.cpu().numpy(), oi)
        np.testing.assert_array_equal(gc.cpu().numpy(), oc)
    with pytest.raises(ValueError):
        ops.query_ball_point_multi(probs + probs[:1])
    with pytest.raises(ValueError):
        ops.query_ball_point_multi([(-1.0, 8, T(y, dev), T(y, dev))])


def test_group_point_multi_equals_separate_calls(ops, dev):
    g = torch.Generator(device="cpu").manual_seed(9)
    probs = []
    for b, n, c, m, ns in ((3, 200, 3, 40, 16), (3, 64, 6, 7, 64), (2, 50, 128, 9, 8), (2, 33, 3, 1, 5)):      # xyz / scalar / 16-byte rows
        probs.append((torch.randn(b, n, c, generator=g).to(dev), torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(dev)))
    for got, (p, i) in zip(ops.group_point_multi(probs), probs):
        assert torch.equal(got, ops.group_point(p, i))


@pytest.mark.parametrize("n,m,k", [(64, 5, 8), (1024, 33, 32), (300, 7, 300), (2048, 3, 64), (5, 2, 9)])
def test_select_top_k_and_knn_point(ops, oracle, ref, dev, n, m, k):
    """select_top_k: ALL n columns (sorted head and the swap-permuted tail) equal to the oracle and to the reference's own
    selection_sort_gpu; knn_point = its first k columns over on-the-fly squared distances.  Coarse lattice: many exact ties."""
    rng = np.random.RandomState(n + m + k)
    dist = (rng.randint(0, 40, (2, m, n)) / 8.0).astype(np.float32)
    wi, wo = oracle.select_top_k(k, dist)
    gi, go = ops.select_top_k(k, T(dist, dev))
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(go.cpu().numpy(), wo)
    ri = torch.zeros((2, m, n), dtype=torch.int32, device=dev)
    ro = torch.zeros((2, m, n), dtype=torch.float32, device=dev)
    vp = ctypes.c_void_p
    dt = T(dist, dev)
    assert ref.ref_selection_sort(2, n, m, k, vp(dt.data_ptr()), vp(ri.data_ptr()), vp(ro.data_ptr())) == 0
    np.testing.assert_array_equal(ri.cpu().numpy(), wi)
    np.testing.assert_array_equal(ro.cpu().numpy(), wo)
    if k <= n:
        x1, x2 = cloud(rng, 2, n, "coarse"), cloud(rng, 2, m, "coarse")
        wv, wx = oracle.knn_point(k, x1, x2)
        gv, gx = ops.knn_point(k, T(x1, dev), T(x2, dev))
        np.testing.assert_array_equal(gx.cpu().numpy(), wx)
        np.testing.assert_array_equal(gv.cpu().numpy(), wv)
        d = ((x1[:, None, :, :] - x2[:, :, None, :]) ** 2).sum(-1)
        np.testing.assert_allclose(np.sort(d, axis=2)[:, :, :k], gv.cpu().numpy(), rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        ops.select_top_k(0, T(dist, dev))


def test_ball_query_empty_ball(ops, oracle, dev):
    x = np.zeros((1, 10, 3), np.float32)
    q = np.full((1, 4, 3), 9.0, np.float32)
    gidx, gcnt = ops.query_ball_point(0.5, 8, T(x, dev), T(q, dev))
    assert int(gcnt.abs().sum()) == 0 and int(gidx.abs().sum()) == 0
    widx, wcnt = oracle.query_ball_point(0.5, 8, x, q)   # oracle leaves zero-initialised slots untouched
    np.testing.assert_array_equal(gidx.cpu().numpy(), widx)


@pytest.mark.parametrize("c", [0, 1, 3, 6, 128, 131])
def test_group_point(ops, oracle, dev, c):
    rng = np.random.RandomState(c)
    pts = rng.randn(3, 200, c).astype(np.float32)
    idx = rng.randint(0, 200, (3, 40, 16)).astype(np.int32)
    got = ops.group_point(T(pts, dev), T(idx, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.group_point(pts, idx))


@pytest.mark.parametrize("n,m", [(128, 1), (512, 128), (1024, 512), (2048, 512), (77, 2), (300, 2500)])
@pytest.mark.parametrize("kind", ["uniform", "grid", "coarse"])
def test_three_nn(ops, oracle, dev, n, m, kind):
    rng = np.random.RandomState(n + 3 * m)
    x1, x2 = cloud(rng, 2, n, kind), cloud(rng, 2, m, kind)
    wd, wi = oracle.three_nn(x1, x2)
    gd, gi = ops.three_nn(T(x1, dev), T(x2, dev))
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(gd.cpu().numpy(), wd)   # includes +inf slots when m < 3


@pytest.mark.parametrize("n,m", [(128, 1), (77, 2), (512, 128), (1024, 512), (2048, 512), (300, 2500)])
@pytest.mark.parametrize("kind", ["uniform", "grid", "coarse", "tiled"])
def test_three_nn_and_interpolate_vs_reference_loops(ops, oracle, dev, n, m, kind):
    """HIP == the reference's own threenn_cpu / threeinterpolate_cpu (tf_interpolate.cpp:60-127 compiled where it lies
    into oracle/_ref/libancsh_ref_interp.so): distances, indices (incl. +inf / index-0 slots for m < 3) and interpolated
    rows bit for bit."""
    if not oracle.have_ref_interp():
        pytest.skip("oracle/_ref/libancsh_ref_interp.so not built")
    from articulated_pose_amd.tf_ops.tf_interpolate import three_weights
    rng = np.random.RandomState(n + 5 * m)
    x1, x2 = cloud(rng, 2, n, kind), cloud(rng, 2, m, kind)
    if kind == "tiled":
        x1[:, : min(n, m)] = x2[:, : min(n, m)]
    rd, ri = oracle.ref_three_nn(x1, x2)
    gd, gi = ops.three_nn(T(x1, dev), T(x2, dev))
    np.testing.assert_array_equal(gi.cpu().numpy(), ri)
    np.testing.assert_array_equal(gd.cpu().numpy(), rd)
    w = three_weights(gd)
    pts = rng.randn(2, m, 40).astype(np.float32)
    got = ops.three_interpolate(T(pts, dev), gi, w).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.ref_three_interpolate(pts, ri, w.cpu().numpy()))


def test_three_weights_and_interpolate(ops, oracle, dev):
    from articulated_pose_amd.tf_ops.tf_interpolate import three_weights
    rng = np.random.RandomState(5)
    x1, x2 = cloud(rng, 2, 512), cloud(rng, 2, 128)
    x1[0, :5] = x2[0, :5]                       # zero distances -> 1e-10 clamp
    d, i = oracle.three_nn(x1, x2)
    w = oracle.three_weights(d)
    gw = three_weights(T(d, dev)).cpu().numpy()
    np.testing.assert_array_equal(gw, w)
    pts = rng.randn(2, 128, 256).astype(np.float32)
    got = ops.three_interpolate(T(pts, dev), T(i, dev), T(w, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.three_interpolate(pts, i, w))
    # m = 1 (FP1): weights must be exactly (1,0,0)
    d1, i1 = oracle.three_nn(x1, x2[:, :1])
    w1 = three_weights(T(d1, dev)).cpu().numpy()
    np.testing.assert_array_equal(w1, np.broadcast_to(np.array([1, 0, 0], np.float32), w1.shape))


# ---- the reference's own CUDA sources, compiled by hipcc for gfx950 (oracle/_ref) -------------
# Inputs on a 2^-8 lattice: every squared distance is exact in float32, so the result does not
# depend on the compiler's FMA contraction; exact ties exercise the reference's tie-breaking.
@pytest.mark.parametrize("kind", ["grid", "coarse", "tiled"])
@pytest.mark.parametrize("n,m", [(1024, 512), (512, 128), (2048, 512), (3000, 100), (3500, 64)])
def test_fps_vs_reference_kernel(ops, oracle, ref, dev, kind, n, m):
    rng = np.random.RandomState(n + m + 11)
    x = cloud(rng, 34, n, kind)     # 34 clouds > 32 blocks: exercises the reference's grid-stride loop
    if kind == "tiled":
        x = (np.round(x * 256) / 256).astype(np.float32)
    xt = T(x, dev)
    out = torch.zeros((34, m), dtype=torch.int32, device=dev)
    temp = torch.zeros((32, n), dtype=torch.float32, device=dev)
    rc = ref.ref_farthest_point_sample(34, n, m, ctypes.c_void_p(xt.data_ptr()), ctypes.c_void_p(temp.data_ptr()),
                                       ctypes.c_void_p(out.data_ptr()))
    assert rc == 0
    want = out.cpu().numpy()
    np.testing.assert_array_equal(ops.farthest_point_sample(m, xt).cpu().numpy(), want)
    np.testing.assert_array_equal(oracle.farthest_point_sample(m, x[:4]), want[:4])


@pytest.mark.parametrize("kind", ["grid", "coarse"])
@pytest.mark.parametrize("n,m,r,ns", [(1024, 512, 0.2, 64), (512, 128, 0.4, 64), (2048, 300, 0.25, 32)])
def test_ball_query_group_vs_reference_kernel(ops, oracle, ref, dev, kind, n, m, r, ns):
    rng = np.random.RandomState(n + m + 13)
    x = cloud(rng, 5, n, kind)
    q = oracle.gather_point(x, oracle.farthest_point_sample(m, x))
    xt, qt = T(x, dev), T(q, dev)
    idx = torch.zeros((5, m, ns), dtype=torch.int32, device=dev)
    cnt = torch.zeros((5, m), dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p
    assert ref.ref_query_ball_point(5, n, m, ctypes.c_float(r), ns, vp(xt.data_ptr()), vp(qt.data_ptr()),
                                    vp(idx.data_ptr()), vp(cnt.data_ptr())) == 0
    gidx, gcnt = ops.query_ball_point(r, ns, xt, qt)
    np.testing.assert_array_equal(gidx.cpu().numpy(), idx.cpu().numpy())
    np.testing.assert_array_equal(gcnt.cpu().numpy(), cnt.cpu().numpy())
    widx, wcnt = oracle.query_ball_point(r, ns, x, q)
    np.testing.assert_array_equal(widx, idx.cpu().numpy())
    feats = torch.randn((5, n, 16), device=dev)
    out = torch.zeros((5, m, ns, 16), device=dev)
    assert ref.ref_group_point(5, n, 16, m, ns, vp(feats.data_ptr()), vp(idx.data_ptr()), vp(out.data_ptr())) == 0
    assert torch.equal(ops.group_point(feats, idx), out)


def test_argument_errors(ops, dev):
    x = torch.zeros((2, 16, 3), device=dev)
    with pytest.raises(ValueError):
        ops.farthest_point_sample(0, x)
    with pytest.raises(ValueError):
        ops.farthest_point_sample(4, x[..., :2])
    with pytest.raises(ValueError):
        ops.query_ball_point(-1.0, 4, x, x)
    with pytest.raises(ValueError):
        ops.query_ball_point(0.1, 0, x, x)
    with pytest.raises(ValueError):
        ops.group_point(x, torch.zeros((3, 4, 4), dtype=torch.int32, device=dev))
    with pytest.raises(RuntimeError):
        ops.farthest_point_sample(4, x.cpu())


@pytest.mark.parametrize("b,m,c2,n,c1,ld", [(3, 128, 256, 512, 128, 384), (2, 512, 128, 1024, 3, 132), (2, 7, 8, 33, 0, 8), (1, 1, 1024, 128, 256, 1280),
                                            (8, 128, 256, 512, 128, 384), (16, 16, 8, 512, 3, 12), (24, 512, 128, 1024, 3, 132)])   # multiples of 8 clouds: the XCD-aware block map
def test_fp_interpolate_concat_equals_separate_ops(dev, b, m, c2, n, c1, ld):
    """ancsh_fp_interpolate_concat == three_interpolate into the row + copy of points1 + zero pad, bit for bit."""
    from articulated_pose_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(b * 1000 + n)
    p2 = torch.randn(b, m, c2, generator=g).to(dev)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32).to(dev)
    w = torch.rand(b, n, 3, generator=g).to(dev)
    p1 = torch.randn(b, n, max(c1, 1), generator=g).to(dev)[..., :c1].contiguous()
    want = torch.zeros(b, n, ld, device=dev)
    _lib.call("ancsh_three_interpolate_ex", b, m, c2, n, _lib.ptr(p2), _lib.ptr(idx), _lib.ptr(w), _lib.ptr(want), ld, 0)
    if c1:
        want[..., c2:c2 + c1] = p1
    got = torch.full((b, n, ld), float("nan"), device=dev)
    _lib.call("ancsh_fp_interpolate_concat", b, m, c2, n, _lib.ptr(p2), _lib.ptr(idx), _lib.ptr(w), _lib.ptr(p1) if c1 else None, c1, _lib.ptr(got), ld)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        _lib.call("ancsh_fp_interpolate_concat", b, m, c2, n, _lib.ptr(p2), _lib.ptr(idx), _lib.ptr(w), _lib.ptr(p1), c1, _lib.ptr(got), ld - 4 if c1 else 4)


@pytest.mark.parametrize("n,m,r,ns", [(1024, 512, 0.2, 64), (512, 128, 0.4, 64), (333, 77, 0.3, 16), (100, 10, 0.01, 8), (100, 10, 5.0, 64)])
@pytest.mark.parametrize("center", [False, True])
def test_query_ball_group_xyz_equals_separate_ops(ops, dev, n, m, r, ns, center):
    """The fused launch against query_ball_point + group_point (+ centroid subtraction): identical idx, counts and rows."""
    rng = np.random.RandomState(n * 7 + m)
    x = T(cloud(rng, 3, n, "coarse"), dev)
    q = x[:, :m].contiguous() if m <= n else T(cloud(rng, 3, m, "coarse"), dev)
    idx, cnt = ops.query_ball_point(r, ns, x, q)
    g = ops.group_point(x, idx)
    if center:
        g = g - q[:, :, None, :]
    fi, fc, fg = ops.query_ball_group_xyz(r, ns, x, q, center=center)
    assert torch.equal(fi, idx) and torch.equal(fc, cnt)
    assert torch.equal(fg, g)


@pytest.mark.parametrize("center", [False, True])
def test_query_ball_group_xyz_multi_equals_separate_calls(ops, dev, center):
    """Both SA levels' ball query + xyz grouping (and two odd-shaped problems) in ONE launch: every output equal to the
    single-problem entry, which test_query_ball_group_xyz_equals_separate_ops ties to the two reference operators."""
    rng = np.random.RandomState(11)
    x = T(cloud(rng, 4, 1024, "coarse"), dev)
    l1 = x[:, :512].contiguous()
    l2 = l1[:, :128].contiguous()
    y = T(cloud(rng, 2, 333, "uniform"), dev)
    probs = [(0.2, 64, x, l1), (0.4, 64, l1, l2), (0.3, 16, y, y[:, :77].contiguous()), (0.01, 8, y, y[:, :5].contiguous())]
    got = ops.query_ball_group_xyz_multi(probs, center=center)
    for (r, ns, a, q), (gi, gc, gg) in zip(probs, got):
        wi, wc, wg = ops.query_ball_group_xyz(r, ns, a, q, center=center)
        assert torch.equal(gi, wi) and torch.equal(gc, wc) and torch.equal(gg, wg)
    with pytest.raises(ValueError):
        ops.query_ball_group_xyz_multi(probs + probs[:1])


def test_three_nn_weights_one_launch_equals_two(dev):
    """ancsh_three_nn_weights = ancsh_three_nn + ancsh_three_weights, bit for bit (also m < 3: +inf distances)"""
    from articulated_pose_amd.tf_ops import tf_interpolate as ti
    rng = np.random.RandomState(4)
    for b, n, m in ((3, 1024, 512), (2, 777, 128), (1, 64, 2), (2, 50, 1)):
        x1 = torch.from_numpy(rng.rand(b, n, 3).astype(np.float32)).to(dev)
        x2 = torch.from_numpy(rng.rand(b, m, 3).astype(np.float32)).to(dev)
        d, i = ti.three_nn(x1, x2)
        w = ti.three_weights(d)
        d2, i2, w2 = ti.three_nn_weights(x1, x2)
        assert torch.equal(d, d2) and torch.equal(i, i2)
        assert torch.equal(torch.nan_to_num(w, nan=-1.0), torch.nan_to_num(w2, nan=-1.0))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_lane_per_query_ball_query_schedule_equals_the_default(ops, oracle, dev, k, tmp_path):
    """ANCSH_BQ_SCHEDULE=lanes<k> (the opt-in lane = query kernel of csrc/grouping.hip, read once per process) in a SUBPROCESS against
    this process's default schedule: indices, counts and grouped coordinates bit-equal on dense and sparse balls, ragged sizes, the
    fused grouping; ancsh_last_ball_query_schedule() proves which kernel ran -- a request the kernel's LDS / word limits cannot
    serve must fall back to the wave kernel (and say so), not run a kernel the test did not mean."""
    import subprocess
    import sys
    rng = np.random.RandomState(40 + k)
    cases = [(3, 1024, 512, 0.2, 64, "uniform"), (2, 2048, 512, 0.05, 64, "uniform"), (2, 777, 130, 0.3, 16, "grid"), (1, 64, 64, 5.0, 8, "coarse"),
             (1, 20000, 40, 0.2, 32, "uniform")]                     # the last one exceeds every lanes<k> tile: falls back
    arrays = {}
    for i, (b, n, m, r, ns, kind) in enumerate(cases):
        x, q = cloud(rng, b, n, kind), cloud(rng, b, m, kind)
        arrays["x%d" % i], arrays["q%d" % i] = x, q
    np.savez(tmp_path / "in.npz", **arrays)
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
import articulated_pose_amd
from articulated_pose_amd import tf_ops, _lib
d = np.load(%r); out = {}
cases = %r
for i, (b, n, m, r, ns, kind) in enumerate(cases):
    x, q = torch.from_numpy(d['x%%d' %% i]).cuda(), torch.from_numpy(d['q%%d' %% i]).cuda()
    gi, gc = tf_ops.query_ball_point(r, ns, x, q)
    out['sched%%d' %% i] = np.array(_lib.lib().ancsh_last_ball_query_schedule())
    fi, fc, fg = tf_ops.query_ball_group_xyz(r, ns, x, q, center=bool(i & 1))
    out['schedf%%d' %% i] = np.array(_lib.lib().ancsh_last_ball_query_schedule())
    out.update({'gi%%d' %% i: gi.cpu().numpy(), 'gc%%d' %% i: gc.cpu().numpy(), 'fi%%d' %% i: fi.cpu().numpy(), 'fg%%d' %% i: fg.cpu().numpy()})
np.savez(%r, **out)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "in.npz"), cases, str(tmp_path / "out.npz"))
    env = dict(os.environ, ANCSH_BQ_SCHEDULE="lanes%d" % k)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(tmp_path / "out.npz")
    ran = []
    for i, (b, n, m, r_, ns, kind) in enumerate(cases):
        x, q = T(arrays["x%d" % i], dev), T(arrays["q%d" % i], dev)
        wi, wc = ops.query_ball_point(r_, ns, x, q)
        assert ops_lib().ancsh_last_ball_query_schedule() == -1              # this process runs the default
        fi, fc, fg = ops.query_ball_group_xyz(r_, ns, x, q, center=bool(i & 1))
        np.testing.assert_array_equal(got["gi%d" % i], wi.cpu().numpy(), err_msg=str(cases[i]))
        np.testing.assert_array_equal(got["gc%d" % i], wc.cpu().numpy(), err_msg=str(cases[i]))
        np.testing.assert_array_equal(got["fi%d" % i], fi.cpu().numpy(), err_msg=str(cases[i]))
        np.testing.assert_array_equal(got["fg%d" % i], fg.cpu().numpy(), err_msg=str(cases[i]))
        assert int(got["sched%d" % i]) in (k, -1) and int(got["schedf%d" % i]) in (k, -1)
        ran.append(int(got["sched%d" % i]))
    assert ran[0] == k and ran[3] == k, ran          # the network's own shape and a tiny one run the requested kernel
    assert ran[-1] == -1, ran                        # 20000 points do not fit its tile: the fallback is visible


def ops_lib():
    from articulated_pose_amd import _lib
    return _lib.lib()
