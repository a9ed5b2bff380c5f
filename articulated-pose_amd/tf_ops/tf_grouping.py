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


def query_ball_point_multi(problems):
    '''Several independent ball queries in ONE launch.  problems: [(radius, nsample, xyz1, xyz2), ...] (at most 4) ->
    [(idx, pts_cnt), ...], each pair identical to query_ball_point(radius, nsample, xyz1, xyz2).  Used where the operator
    graph has independent ball queries, e.g. layer1 and layer2 of the backbone (layer2 needs the level-1 centroids only).'''
    import ctypes
    if not 1 <= len(problems) <= 4:
        raise ValueError("query_ball_point_multi takes 1..4 problems")
    keep, outs = [], []
    for radius, nsample, xyz1, xyz2 in problems:
        _lib.require_cuda(xyz1, xyz2)
        if not radius > 0:
            raise ValueError("QueryBallPoint expects positive radius")
        if not nsample > 0:
            raise ValueError("QueryBallPoint expects positive nsample")
        if xyz1.dim() != 3 or xyz1.shape[2] != 3:
            raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
        if xyz2.dim() != 3 or xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
            raise ValueError("QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
        cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
        keep.append((b, n, m, float(radius), int(nsample), xyz1, xyz2, idx, cnt))
        outs.append((idx, cnt))
    k = len(keep)
    ints = lambda i: (ctypes.c_int * k)(*[p[i] for p in keep])
    ptrs = lambda i: (ctypes.c_void_p * k)(*[_lib.ptr(p[i]) for p in keep])
    rad = (ctypes.c_float * k)(*[p[3] for p in keep])
    args = [ints(0), ints(1), ints(2), rad, ints(4), ptrs(5), ptrs(6), ptrs(7), ptrs(8)]
    _lib.call("ancsh_query_ball_point_multi", k, *[ctypes.cast(a, ctypes.c_void_p) for a in args])
    return outs


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


def query_ball_group_xyz_multi(problems, center=False):
    '''query_ball_group_xyz for several independent problems in ONE launch.  problems: [(radius, nsample, xyz1, xyz2), ...]
    (at most 4) -> [(idx, pts_cnt, grouped_xyz), ...], each triple identical to query_ball_group_xyz(radius, nsample, xyz1, xyz2).'''
    import ctypes
    if not 1 <= len(problems) <= 4:
        raise ValueError("query_ball_group_xyz_multi takes 1..4 problems")
    keep, outs = [], []
    for radius, nsample, xyz1, xyz2 in problems:
        _lib.require_cuda(xyz1, xyz2)
        if not radius > 0:
            raise ValueError("QueryBallPoint expects positive radius")
        if not nsample > 0:
            raise ValueError("QueryBallPoint expects positive nsample")
        if xyz1.dim() != 3 or xyz1.shape[2] != 3:
            raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
        if xyz2.dim() != 3 or xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
            raise ValueError("QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
        cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
        g = torch.empty((b, m, nsample, 3), dtype=torch.float32, device=xyz1.device)
        keep.append((b, n, m, float(radius), int(nsample), xyz1, xyz2, 1 if center else 0, idx, cnt, g, 3))
        outs.append((idx, cnt, g))
    k = len(keep)
    ints = lambda i: (ctypes.c_int * k)(*[p[i] for p in keep])
    ptrs = lambda i: (ctypes.c_void_p * k)(*[_lib.ptr(p[i]) for p in keep])
    rad = (ctypes.c_float * k)(*[p[3] for p in keep])
    args = [ints(0), ints(1), ints(2), rad, ints(4), ptrs(5), ptrs(6), ints(7), ptrs(8), ptrs(9), ptrs(10), ints(11)]
    _lib.call("ancsh_query_ball_group_xyz_multi", k, *[ctypes.cast(a, ctypes.c_void_p) for a in args])
    return outs


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


def group_point_multi(problems):
    '''Several independent row gathers: problems = [(points, idx), ...] (at most 4) -> [grouped, ...], each identical to
    group_point(points, idx).  The 3-channel problems (grouped xyz of several SA levels) run as ONE launch.'''
    import ctypes
    if not 1 <= len(problems) <= 4:
        raise ValueError("group_point_multi takes 1..4 problems")
    keep, outs = [], []
    for points, idx in problems:
        _lib.require_cuda(points, idx)
        if points.dim() != 3:
            raise ValueError("GroupPoint expects (batch_size, num_points, channel) points shape")
        if idx.dim() != 3 or idx.shape[0] != points.shape[0]:
            raise ValueError("GroupPoint expects (batch_size, npoints, nsample) idx shape")
        points, idx = points.contiguous().float(), idx.contiguous().to(torch.int32)
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
        keep.append((b, n, c, m, ns, points, idx, out))
        outs.append(out)
    k = len(keep)
    ints = lambda i: ctypes.cast((ctypes.c_int * k)(*[p[i] for p in keep]), ctypes.c_void_p)
    ptrs = lambda i: ctypes.cast((ctypes.c_void_p * k)(*[_lib.ptr(p[i]) for p in keep]), ctypes.c_void_p)
    args = [ints(0), ints(1), ints(2), ints(3), ints(4), ptrs(5), ptrs(6), ptrs(7)]
    _lib.call("ancsh_group_point_multi", k, *args)
    return outs


def select_top_k(k, dist):
    '''k SMALLEST entries per row: dist (b, m, n) float32 -> idx (b, m, n) int32 and dist_out (b, m, n) whose first k columns are
    the k smallest values in ascending order (ties: lowest index first) and their positions; the other columns hold the rest in
    the order the reference's selection-sort swaps leave them.  Same contract as the SelectionSort op (tf_grouping.py:22-31,
    tf_grouping.cpp:108-136).'''
    _lib.require_cuda(dist)
    if not k > 0:
        raise ValueError("SelectionSort expects positive k")                     # tf_grouping.cpp:113
    if dist.dim() != 3:
        raise ValueError("SelectionSort expects (b,m,n) dist shape.")            # :118
    dist = dist.contiguous().float()
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    _lib.call("ancsh_selection_sort", b, n, m, int(k), _lib.ptr(dist), _lib.ptr(outi), _lib.ptr(out))
    return outi, out


def knn_point(k, xyz1, xyz2):
    '''k nearest dataset points per query: xyz1 (batch_size, ndataset, c), xyz2 (batch_size, npoint, c) -> val (batch_size, npoint, k)
    squared L2 distances ascending, idx (batch_size, npoint, k) int32 (tf_grouping.py:48-74: pairwise squared distances ->
    select_top_k -> first k columns), in one launch without the (b, m, n) matrix.'''
    _lib.require_cuda(xyz1, xyz2)
    if not k > 0:
        raise ValueError("SelectionSort expects positive k")
    if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.shape[0] != xyz2.shape[0] or xyz1.shape[2] != xyz2.shape[2]:
        raise ValueError("knn_point expects (batch_size, ndataset, c) xyz1 and (batch_size, npoint, c) xyz2")
    xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    val = torch.empty((b, m, k), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, m, k), dtype=torch.int32, device=xyz1.device)
    _lib.call("ancsh_knn_point", b, n, m, c, int(k), _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(val), _lib.ptr(idx))
    return val, idx
