"""Evaluation metrics of the reference, batched on the GPU: the per-part numbers that
evaluation/compute_miou.py, eval_pose_err.py and eval_joint_params.py compute one sample and one part at a time with numpy.

    iou_3d_batch            lib/d3_utils.py:55-69      (HIP kernel ancsh_iou_3d: one workgroup per box pair, 50^3 grid in registers)
    get_3d_bbox             lib/d3_utils.py:14-37
    amodal_boxes            evaluation/compute_miou.py:196-221  (per-part box from predicted NOCS extents, posed by (s, R, t))
    rot_diff_degree_batch   lib/d3_utils.py:144-148
    axis_diff_degree_batch  lib/d3_utils.py:137-142
    dist_between_3d_lines_batch  lib/d3_utils.py:165-174
Float64 throughout, like the reference."""
import math

import torch

from .. import _lib


def get_3d_bbox(scale, shift=0.5):
    """(..., 3) extents -> (..., 8, 3) corners in the reference's order (already transposed to corner-major)."""
    s = torch.as_tensor(scale, dtype=torch.float64)
    sign = torch.tensor([[1, 1, 1], [1, 1, -1], [-1, 1, 1], [-1, 1, -1], [1, -1, 1], [1, -1, -1], [-1, -1, 1], [-1, -1, -1]],
                        dtype=torch.float64, device=s.device)
    return sign * (s[..., None, :] / 2) + shift


def amodal_boxes(scale, s, R, t):
    """Box corners of extent `scale` (..., 3) centred at 0.5 in NOCS, scaled by s (...,), rotated by R (..., 3, 3), moved by t (..., 3):
    bbox * s . R^T + t (compute_miou.py:212-221)."""
    bb = get_3d_bbox(scale) * torch.as_tensor(s, dtype=torch.float64, device=scale.device)[..., None, None]
    return bb @ R.transpose(-1, -2).double() + t[..., None, :].double()


def iou_3d_batch(bbox1, bbox2, nres=50, return_counts=False):
    """bbox1, bbox2: (P, 8, 3) float64 CUDA tensors -> (P,) IoU (and (P, 2) int64 {intersection, union})."""
    b1 = bbox1.contiguous().double()
    b2 = bbox2.contiguous().double()
    _lib.require_cuda(b1)
    if b1.shape != b2.shape or b1.dim() != 3 or b1.shape[1:] != (8, 3):
        raise ValueError("iou_3d_batch expects two (P, 8, 3) corner tensors, got %s and %s" % (tuple(b1.shape), tuple(b2.shape)))
    p = b1.shape[0]
    out = torch.empty(p, dtype=torch.float64, device=b1.device)
    cnt = torch.empty((p, 2), dtype=torch.int64, device=b1.device)
    _lib.call("ancsh_iou_3d", p, int(nres), _lib.ptr(b1), _lib.ptr(b2), _lib.ptr(out), _lib.ptr(cnt))
    return (out, cnt) if return_counts else out


def rot_diff_degree_batch(R1, R2):
    """arccos((tr(R1 R2^T) - 1) / 2) mod 2 pi, in degrees; (..., 3, 3) each."""
    tr = (R1.double() * R2.double()).sum(dim=(-1, -2))              # tr(R1 R2^T) = sum_ij R1_ij R2_ij
    return torch.remainder(torch.arccos((tr - 1) / 2), 2 * math.pi) / math.pi * 180


def axis_diff_degree_batch(v1, v2):
    a, b = v1.double(), v2.double()
    r = torch.arccos((a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))) * 180 / math.pi
    return torch.minimum(r, 180 - r)


def dist_between_3d_lines_batch(p1, e1, p2, e2):
    orth = torch.cross(e1.double(), e2.double(), dim=-1)
    return ((orth * (p1.double() - p2.double())).sum(-1) / orth.norm(dim=-1)).abs()
