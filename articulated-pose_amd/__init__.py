"""articulated-pose_amd: MI355X-native ANCSH inference + pose-fit hot path.

Python host code (PyTorch-ROCm for device memory / streams / torch.distributed) over
hand-written gfx950 HIP kernels behind a C ABI (include/ancsh_hip.h, libancsh_hip.so).
Module names mirror the reference tree so its call sites drop in:
    tf_ops.tf_sampling / tf_grouping / tf_interpolate   <- pointnet_plusplus/utils/tf_ops/*/tf_*.py
    pointnet_util, tf_util, architectures               <- pointnet_plusplus/{utils/,}*.py
    architecture, network, prediction_io                <- lib/*.py
    pose.*                                              <- evaluation/parallel_ancsh_pose.py, lib/d3_utils.py, lib/aligning.py
"""
import os as _os

# AncshPipeline keeps several batches in flight on separate HIP streams; each needs its own hardware queue (the runtime's
# default of 4 makes batches wait behind each other's long pose kernels).  Read by the HIP runtime at initialisation.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

__version__ = "0.1.0"
