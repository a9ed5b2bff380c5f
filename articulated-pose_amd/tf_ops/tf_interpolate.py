"""Drop-in for pointnet_plusplus/utils/tf_ops/3d_interpolation/tf_interpolate.py on torch.Tensors.
In the reference both ops are host-only (DEVICE_CPU); here they are gfx950 kernels."""
import torch

from .. import _lib


def three_nn(xyz1, xyz2):
    '''For every point of xyz1 (b, n, 3) the three nearest points of xyz2 (b, m, 3).
    Returns (dist, idx), both (b, n, 3): SQUARED distances in ascending order (float32; +inf where m < 3) and the matching
    int32 indices into xyz2; equal distances keep the lower index.  Same contract as the reference's ThreeNN op
    (tf_interpolate.py:8-17, tf_interpolate.cpp:157-187), which runs on the host.'''
    _lib.require_cuda(xyz1, xyz2)
    if xyz1.dim() != 3 or xyz1.shape[2] != 3:
        raise ValueError("ThreeNN expects (b,n,3) xyz1 shape")     # tf_interpolate.cpp:163
    if xyz2.dim() != 3 or xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
        raise ValueError("ThreeNN expects (b,m,3) xyz2 shape")     # :168
    xyz1 = xyz1.contiguous().float()
    xyz2 = xyz2.contiguous().float()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    _lib.call("ancsh_three_nn", b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist), _lib.ptr(idx))
    return dist, idx


def three_nn_weights(xyz1, xyz2):
    """three_nn followed by three_weights in ONE launch: (dist, idx, weight), each (b, n, 3); same values as the two ops."""
    _lib.require_cuda(xyz1, xyz2)
    if xyz1.dim() != 3 or xyz1.shape[2] != 3:
        raise ValueError("ThreeNN expects (b,n,3) xyz1 shape")
    if xyz2.dim() != 3 or xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
        raise ValueError("ThreeNN expects (b,m,3) xyz2 shape")
    xyz1 = xyz1.contiguous().float()
    xyz2 = xyz2.contiguous().float()
    b, n, _ = xyz1.shape
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    weight = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    _lib.call("ancsh_three_nn_weights", b, n, xyz2.shape[1], _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist), _lib.ptr(idx), _lib.ptr(weight))
    return dist, idx, weight


def three_weights(dist):
    """pointnet_util.py:219-222 as one kernel: (1/max(d,1e-10)) / sum(1/max(d,1e-10))."""
    _lib.require_cuda(dist)
    dist = dist.contiguous().float()
    w = torch.empty_like(dist)
    _lib.call("ancsh_three_weights", dist.numel() // 3, _lib.ptr(dist), _lib.ptr(w))
    return w


def three_interpolate(points, idx, weight):
    '''Weighted sum of three feature rows per target point: out[b, j] = sum_k weight[b, j, k] * points[b, idx[b, j, k]].
    points (b, m, c) float32, idx / weight (b, n, 3) -> out (b, n, c).  Same contract as the reference's ThreeInterpolate op
    (tf_interpolate.py:19-28, tf_interpolate.cpp:191-222).'''
    _lib.require_cuda(points, idx, weight)
    if points.dim() != 3:
        raise ValueError("ThreeInterpolate expects (b,m,c) points shape")                 # :197
    b, m, c = points.shape
    if idx.dim() != 3 or idx.shape[0] != b or idx.shape[2] != 3:
        raise ValueError("ThreeInterpolate expects (b,n,3) idx shape")                    # :203
    if weight.dim() != 3 or tuple(weight.shape) != tuple(idx.shape):
        raise ValueError("ThreeInterpolate expects (b,n,3) weight shape")                 # :206
    points = points.contiguous().float()
    idx = idx.contiguous().to(torch.int32)
    weight = weight.contiguous().float()
    n = idx.shape[1]
    out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
    _lib.call("ancsh_three_interpolate", b, m, c, n, _lib.ptr(points), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(out))
    return out
