"""Drop-in for pointnet_plusplus/utils/tf_ops/grouping/tf_grouping.py on torch.Tensors (MI355X)."""
import torch

from .. import _lib


def query_ball_point(radius, nsample, xyz1, xyz2):
    '''For each query xyz2[b, j] the first `nsample` points of xyz1[b] (ascending index) closer than `radius`.
    xyz1 (B, n, 3), xyz2 (B, m, 3) -> idx (B, m, nsample) int32, slots past the hit count repeating the first hit, and
    pts_cnt (B, m) int32 = min(hits, nsample).  Same contract as the QueryBallPoint op (tf_grouping.py:8-21,
    tf_grouping.cpp:67-106).'''
    _lib.require_cuda(xyz1, xyz2)
    if not radius > 0:
        raise ValueError("QueryBallPoint expects positive radius")     # tf_grouping.cpp:71
    if not nsample > 0:
        raise ValueError("QueryBallPoint expects positive nsample")    # tf_grouping.cpp:74
    if xyz1.dim() != 3 or xyz1.shape[2] != 3:
        raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")   # :79
    if xyz2.dim() != 3 or xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
        raise ValueError("QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")     # :84
    xyz1 = xyz1.contiguous().float()
    xyz2 = xyz2.contiguous().float()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    _lib.call("ancsh_query_ball_point", b, n, m, float(radius), int(nsample), _lib.ptr(xyz1), _lib.ptr(xyz2),
              _lib.ptr(idx), _lib.ptr(cnt))
    return idx, cnt


def query_ball_group_xyz(radius, nsample, xyz1, xyz2, center=False):
    '''query_ball_point + group_point(xyz1, idx) [- xyz2] in ONE launch: the first two ops of sample_and_group
    (pointnet_util.py:47-49).  Returns idx, pts_cnt as query_ball_point and grouped_xyz (batch_size, npoint, nsample, 3),
    the same values the two separate ops give.'''
    _lib.require_cuda(xyz1, xyz2)
    if not radius > 0:
        raise ValueError("QueryBallPoint expects positive radius")
    if not nsample > 0:
        raise ValueError("QueryBallPoint expects positive nsample")
    if xyz1.dim() != 3 or xyz1.shape[2] != 3:
        raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
    if xyz2.dim() != 3 or xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
        raise ValueError("QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
    xyz1 = xyz1.contiguous().float()
    xyz2 = xyz2.contiguous().float()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    g = torch.empty((b, m, nsample, 3), dtype=torch.float32, device=xyz1.device)
    _lib.call("ancsh_query_ball_group_xyz", b, n, m, float(radius), int(nsample), _lib.ptr(xyz1), _lib.ptr(xyz2), 1 if center else 0,
              _lib.ptr(idx), _lib.ptr(cnt), _lib.ptr(g), 3)
    return idx, cnt, g


def group_point(points, idx):
    '''Row gather: out[b, j, s, :] = points[b, idx[b, j, s], :]; points (B, n, C) float32, idx (B, m, nsample) int32 ->
    (B, m, nsample, C).  Same contract as the GroupPoint op (tf_grouping.py:33-41, tf_grouping.cpp:143-171).'''
    _lib.require_cuda(points, idx)
    if points.dim() != 3:
        raise ValueError("GroupPoint expects (batch_size, num_points, channel) points shape")   # :149
    if idx.dim() != 3 or idx.shape[0] != points.shape[0]:
        raise ValueError("GroupPoint expects (batch_size, npoints, nsample) idx shape")         # :155
    points = points.contiguous().float()
    idx = idx.contiguous().to(torch.int32)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
    if c > 0:
        _lib.call("ancsh_group_point", b, n, c, m, ns, _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out))
    return out


def select_top_k(k, dist):
    raise NotImplementedError("select_top_k (kNN grouping) is outside the ANCSH inference path: "
                              "knn=False everywhere (pointnet_util.py:94, architectures.py:62-75)")


def knn_point(k, xyz1, xyz2):
    raise NotImplementedError("knn_point is outside the ANCSH inference path (knn=False everywhere)")
