"""Host-side helpers with the names of lib/d3_utils.py that the pose driver needs for its error
bookkeeping (rot_diff_degree :144-148).  The numerically heavy functions of that file (rotate_pts,
scale_pts, transform_pts, rotate_points_with_rotvec) run inside the HIP kernels (csrc/pose_math.h)."""
import numpy as np


def rot_diff_rad(rot1, rot2):
    return np.arccos((np.trace(np.matmul(rot1, rot2.T)) - 1) / 2) % (2 * np.pi)


def rot_diff_degree(rot1, rot2):
    return rot_diff_rad(rot1, rot2) / np.pi * 180
