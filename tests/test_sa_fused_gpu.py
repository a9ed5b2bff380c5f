"""GPU: the fused set-abstraction kernels (csrc/sa_fused.hip, csrc/sa_regchain.h) on RAGGED launches -- neighbourhood counts that are
not a multiple of the four a workgroup takes (dead waves), cloud counts that do not divide by the eight XCDs (the identity
workgroup map), clouds of a few points (every neighbourhood padded with its first index) -- against the op-by-op path of the same module
(ball query -> group -> three ancsh_conv1x1 -> max: each operator pinned against the CPU oracle elsewhere), bit for bit; and the grouped
launch (two networks per launch) against two plain launches."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _weights(rng, cin, mlp, scope):
    w = {}
    for i, c in enumerate(mlp):
        full = "%s/conv%d" % (scope, i)
        w[full + "/weights"] = (rng.randn(1, 1, cin, c) / np.sqrt(cin)).astype(np.float32)
        w[full + "/biases"] = (rng.randn(c) * 0.1).astype(np.float32)
        w[full + "/bn/beta"] = (rng.randn(c) * 0.1).astype(np.float32)
        w[full + "/bn/gamma"] = (rng.uniform(0.5, 1.5, c) * rng.choice([1.0, 1.0, 1.0, -1.0], c)).astype(np.float32)
        w[full + "/bn/moving_mean"] = (rng.randn(c) * 0.1).astype(np.float32)
        w[full + "/bn/moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        cin = c
    return w


@pytest.mark.parametrize("B,n,m,feat", [(1, 100, 5, 0), (3, 777, 13, 0), (9, 1024, 512, 0), (2, 64, 64, 0), (1, 3, 1, 0), (5, 300, 7, 128),
                                        (3, 512, 128, 128), (8, 1024, 30, 0), (16, 256, 6, 128)])
def test_fused_sa_levels_on_ragged_launches(dev, B, n, m, feat):
    from articulated_pose_amd import pointnet_util, tf_util
    rng = np.random.RandomState(B * 1000 + n + m)
    mlp = [128, 128, 256] if feat else [64, 64, 128]
    w = _weights(rng, 3 + feat, mlp, "lvl")
    xyz = torch.from_numpy(rng.uniform(-0.5, 0.5, (B, n, 3)).astype(np.float32)).to(dev)
    pts = torch.from_numpy(rng.randn(B, n, feat).astype(np.float32)).to(dev) if feat else xyz[:, :, 3:3]
    outs = []
    for fused in (True, False):
        tf_util.set_variables(w)
        try:
            pointnet_util.FUSED_SA = fused
            new_xyz, new_pts, idx = pointnet_util.pointnet_sa_module(xyz, pts, m, 0.3, 64, mlp, None, False, False, None, "lvl")
        finally:
            pointnet_util.FUSED_SA = True
        outs.append((new_xyz, new_pts, idx))
    for a, b in zip(outs[0], outs[1]):
        assert a.shape == b.shape and torch.equal(a, b), (B, n, m, feat)
    assert tuple(outs[0][1].shape) == (B, m, mlp[2]) and torch.isfinite(outs[0][1]).all()


@pytest.mark.parametrize("G,B,n,m", [(2, 3, 200, 11), (3, 8, 512, 64), (4, 1, 64, 2)])
def test_grouped_sa1_equals_plain_launches(dev, G, B, n, m):
    from articulated_pose_amd import _lib, tf_ops
    from articulated_pose_amd.tf_ops.tf_sampling import farthest_point_sample_gather
    rng = np.random.RandomState(G * 100 + B)
    xyz = torch.from_numpy(rng.uniform(-0.5, 0.5, (B, n, 3)).astype(np.float32)).to(dev)
    _, new_xyz = farthest_point_sample_gather(m, xyz)
    idx, _ = tf_ops.query_ball_point(0.3, 64, xyz, new_xyz)
    mlp = (64, 64, 128)
    params = []
    for g in range(G):
        cin, layer = 3, []
        for c in mlp:
            wt = torch.from_numpy((rng.randn(cin, c) / np.sqrt(cin)).astype(np.float32)).to(dev)
            pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(cin, c), device=dev)
            _lib.call("ancsh_sa_pack_weights", cin, c, _lib.ptr(wt), _lib.ptr(pk))
            layer += [pk] + [torch.from_numpy(rng.randn(c).astype(np.float32) * s + o).to(dev) for s, o in ((0.1, 0.0), (0.3, 1.0), (0.1, 0.0))]
            cin = c
        params.append(layer)
    flat = [t for layer in params for t in layer]
    ptrs = (ctypes.c_void_p * (12 * G))(*[_lib.ptr(t) for t in flat])
    out_g = torch.empty((G * B, m, 128), device=dev)
    _lib.call("ancsh_sa_module_fused_grouped", G, B, n, m, 64, 0, *mlp, _lib.ptr(xyz), None, _lib.ptr(new_xyz), _lib.ptr(idx),
              ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out_g))
    for g in range(G):
        p1 = (ctypes.c_void_p * 12)(*[_lib.ptr(t) for t in params[g]])
        out = torch.empty((B, m, 128), device=dev)
        _lib.call("ancsh_sa_module_fused", B, n, m, 64, 0, *mlp, _lib.ptr(xyz), None, _lib.ptr(new_xyz), _lib.ptr(idx),
                  ctypes.cast(p1, ctypes.c_void_p), _lib.ptr(out))
        assert torch.equal(out_g[g * B:(g + 1) * B], out), (G, B, n, m, g)
