"""articulated-pose_amd: MI355X-native ANCSH inference + pose-fit hot path.

Python host code (PyTorch-ROCm for device memory / streams / torch.distributed) over
hand-written gfx950 HIP kernels behind a C ABI (include/ancsh_hip.h, libancsh_hip.so).
Module names mirror the reference tree so its call sites drop in:
    tf_ops.tf_sampling / tf_grouping / tf_interpolate   <- pointnet_plusplus/utils/tf_ops/*/tf_*.py
    pointnet_util, tf_util, architectures               <- pointnet_plusplus/{utils/,}*.py
    architecture, network, prediction_io                <- lib/*.py
    pose.*                                              <- evaluation/parallel_ancsh_pose.py, lib/d3_utils.py, lib/aligning.py
"""
__version__ = "0.1.0"

# Importing this package has NO side effects on the process (until round 4 it exported GPU_MAX_HW_QUEUES=32 here).  The one runtime
# setting the throughput deployment needs -- one hardware queue per batch in flight -- is requested by pipeline.AncshPipeline when it is
# built with more than four slots (see pipeline.ensure_hardware_queues), and a host that initialises HIP first sets it itself
# (include/ancsh_hip.h "THROUGHPUT NOTE"; bench.py does so at its top).
