"""Drop-in for pointnet_plusplus/utils/tf_ops/sampling/tf_sampling.py (same names, argument
order and return shapes) on torch.Tensors resident on the MI355X."""
import torch

from .. import _lib


def _check_xyz(name, t, what):
    if t.dim() != 3 or t.shape[2] != 3:
        raise ValueError(f"{name} expects {what} shape")   # tf_sampling.cpp:105,131


def prob_sample(inp, inpr):
    '''Inverse-CDF sampling: inp (batch_size, ncategory) float32 weights, inpr (batch_size, npoints) float32 uniform randoms ->
    (batch_size, npoints) int32 category indices.  Same contract as the ProbSample op (tf_sampling.py:13-21,
    tf_sampling.cpp:66-92).'''
    _lib.require_cuda(inp, inpr)
    if inp.dim() != 2:
        raise ValueError("ProbSample expects (batch_size,num_choices) inp shape")          # tf_sampling.cpp:76
    if inpr.dim() != 2 or inpr.shape[0] != inp.shape[0]:
        raise ValueError("ProbSample expects (batch_size,num_points) inpr shape")         # tf_sampling.cpp:79
    inp, inpr = inp.contiguous().float(), inpr.contiguous().float()
    b, n = inp.shape
    m = inpr.shape[1]
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    _lib.call("ancsh_prob_sample", b, n, m, _lib.ptr(inp), _lib.ptr(inpr), _lib.ptr(temp), _lib.ptr(out))
    return out


def _fps_scratch(b, n, device):
    """The reference op allocates a 32*n-float `temp` (tf_sampling.cpp:115).  Here clouds up to 8192 points keep their running
    minimum distances in registers (no scratch); larger clouds use the large-cloud kernel, which needs b*n floats."""
    return torch.empty((b, n), dtype=torch.float32, device=device) if n > 8192 else None


def farthest_point_sample(npoint, inp):
    '''Iterative farthest-point picks: inp (B, n, 3) float32 -> (B, npoint) int32 indices, starting from point 0 and
    breaking distance ties exactly as the reference's 512-thread kernel does.  Same contract as the FarthestPointSample op
    (tf_sampling.py:48-57, tf_sampling.cpp:95-123).'''
    _lib.require_cuda(inp)
    _check_xyz("FarthestPointSample", inp, "(batch_size,num_points,3) inp")
    if npoint <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")   # tf_sampling.cpp:99
    inp = inp.contiguous().float()
    b, n, _ = inp.shape
    out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    _lib.call("ancsh_farthest_point_sample", b, n, npoint, _lib.ptr(inp), _lib.ptr(_fps_scratch(b, n, inp.device)), _lib.ptr(out))
    return out


def farthest_point_sample_gather(npoint, inp):
    """Fused farthest_point_sample + gather_point (pointnet_util.py:47): returns (idx, new_xyz)."""
    _lib.require_cuda(inp)
    _check_xyz("FarthestPointSample", inp, "(batch_size,num_points,3) inp")
    if npoint <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")
    inp = inp.contiguous().float()
    b, n, _ = inp.shape
    idx = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    xyz = torch.empty((b, npoint, 3), dtype=torch.float32, device=inp.device)
    _lib.call("ancsh_farthest_point_sample_gather", b, n, npoint, _lib.ptr(inp), _lib.ptr(_fps_scratch(b, n, inp.device)),
              _lib.ptr(idx), _lib.ptr(xyz))
    return idx, xyz


def gather_point(inp, idx):
    '''out[b, j, :] = inp[b, idx[b, j], :]: inp (B, n, 3) float32, idx (B, m) int32 -> (B, m, 3).  Same contract as the
    GatherPoint op (tf_sampling.py:29-37, tf_sampling.cpp:126-148).'''
    _lib.require_cuda(inp, idx)
    _check_xyz("GatherPoint", inp, "(batch_size,num_points,3) inp")
    if idx.dim() != 2 or idx.shape[0] != inp.shape[0]:
        raise ValueError("GatherPoint expects (batch_size,num_result) idx shape")   # tf_sampling.cpp:135
    inp = inp.contiguous().float()
    idx = idx.contiguous().to(torch.int32)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
    _lib.call("ancsh_gather_point", b, n, m, _lib.ptr(inp), _lib.ptr(idx), _lib.ptr(out))
    return out
