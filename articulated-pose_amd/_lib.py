"""ctypes binding of libancsh_hip.so (C ABI declared in include/ancsh_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this module
raises.  It never imports anything from oracle/.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANCSH_HIP_LIB") or os.path.join(_HERE, "libancsh_hip.so")   # override: an experimental build

_c_int, _c_long, _c_float, _vp = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p

# name -> argtypes (every function returns int status except the two diagnostics)
SIGNATURES = {
    "ancsh_farthest_point_sample": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "ancsh_farthest_point_sample_gather": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_prob_sample": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_gather_point": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "ancsh_query_ball_point": [_c_int, _c_int, _c_int, _c_float, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_query_ball_point_multi": [_c_int] + [_vp] * 10,
    "ancsh_group_point": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "ancsh_group_point_multi": [_c_int] + [_vp] * 9,
    "ancsh_selection_sort": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "ancsh_knn_point": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_group_point_ex": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _vp],
    "ancsh_three_nn": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_three_weights": [_c_int, _vp, _vp, _vp],
    "ancsh_three_nn_weights": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ancsh_three_interpolate": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_three_interpolate_ex": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _vp],
    "ancsh_conv1x1": [_c_long, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _vp],
    "ancsh_conv1x1_ex": [_c_long, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _vp],
    "ancsh_conv1x1_packed": [_c_long, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _vp],
    "ancsh_conv1x1_packed_grouped": [_c_int, _c_long, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _vp],
    "ancsh_conv1x1_grouped": [_c_int, _c_long, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _vp],
    "ancsh_sa_module_fused_grouped": [_c_int] * 9 + [_vp] * 4 + [_vp, _vp, _vp],
    "ancsh_sa_module_fused_partial_grouped": [_c_int] * 8 + [_vp] * 7,
    "ancsh_fp_interpolate_concat_ex": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _vp],
    "ancsh_sa_pack_weights_bf16x3": [_c_int, _c_int, _vp, _vp, _vp],
    "ancsh_sa_module_fused_bf16x3": [_c_int] * 8 + [_vp] * 4 + [_vp, _vp, _vp],
    "ancsh_sa_module_fused_partial_bf16x3": [_c_int] * 7 + [_vp] * 7,
    "ancsh_sa_module_fused_bf16x3_grouped": [_c_int] * 9 + [_vp] * 4 + [_vp, _vp, _vp],
    "ancsh_sa_module_fused_partial_bf16x3_grouped": [_c_int] * 8 + [_vp] * 7,
    "ancsh_sa_pack_weights_f16x2": [_c_int, _c_int, _vp, _vp, _vp],
    "ancsh_sa_module_fused_f16x2_grouped": [_c_int] * 9 + [_vp] * 4 + [_vp, _vp, _vp],
    "ancsh_sa_module_fused_partial_f16x2_grouped": [_c_int] * 8 + [_vp] * 7,
    "ancsh_mlp_chain_grouped_fp_f16x2": [_c_int] * 5 + [_vp] * 7 + [_vp],
    "ancsh_sa3_chain_grouped_bf16x3": [_c_int] * 7 + [_vp] * 4 + [_vp],
    "ancsh_sa3_chain_grouped_f16x2": [_c_int] * 7 + [_vp] * 4 + [_vp],
    "ancsh_fp1_chain_grouped_bf16x3": [_c_int] * 6 + [_vp] * 4 + [_vp],
    "ancsh_fp1_chain_grouped_f16x2": [_c_int] * 6 + [_vp] * 4 + [_vp],
    "ancsh_fp2_chain_grouped_bf16x3": [_c_int] * 8 + [_vp] * 6 + [_vp],
    "ancsh_fp2_chain_grouped_f16x2": [_c_int] * 8 + [_vp] * 6 + [_vp],
    "ancsh_iou_3d": [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_joint_params": [_c_int] * 5 + [_vp] * 9 + [_vp],
    "ancsh_part_extents": [_c_int] * 4 + [_vp] * 3 + [_c_int] + [_vp] * 4 + [_vp],
    "ancsh_fp_interpolate_concat": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _vp],
    "ancsh_query_ball_group_xyz": [_c_int, _c_int, _c_int, _c_float, _c_int, _vp, _vp, _c_int, _vp, _vp, _vp, _c_int, _vp],
    "ancsh_query_ball_group_xyz_multi": [_c_int] + [_vp] * 13,
    "ancsh_group_max": [_c_long, _c_int, _c_int, _vp, _vp, _vp],
    "ancsh_sa_module_fused": [_c_int] * 8 + [_vp] * 4 + [_vp, _vp, _vp],
    "ancsh_sa_module_fused_partial": [_c_int] * 7 + [_vp] * 7,
    "ancsh_sa_pack_weights": [_c_int, _c_int, _vp, _vp, _vp],
    "ancsh_mlp_chain": [_c_long, _c_int, _vp, _c_int, _c_int, _vp, _vp, _vp],
    "ancsh_mlp_chain_grouped": [_c_int, _c_long, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_sa3_chain_grouped": [_c_int] * 7 + [_vp] * 4 + [_vp],
    "ancsh_fp_single_source_init": [_c_int] * 5 + [_vp] * 3 + [_vp],
    "ancsh_fp1_chain_grouped": [_c_int] * 6 + [_vp] * 4 + [_vp],
    "ancsh_fp2_chain_grouped": [_c_int] * 8 + [_vp] * 6 + [_vp],
    "ancsh_mlp_chain_grouped_fp": [_c_int] * 5 + [_vp] * 8 + [_vp],
    "ancsh_mlp_chain_grouped_fp_bf16x3": [_c_int] * 5 + [_vp] * 7 + [_vp],
    "ancsh_head_activations": [_c_long, _c_int, _c_int, _vp, _c_int] + [_vp] * 10 + [_vp],
    "ancsh_pose_partition": [_c_int, _c_int, _c_int] + [_vp] * 11 + [_vp],
    "ancsh_pose_poison_records": [_c_int, _c_int, _c_int] + [_vp] * 5 + [_vp],
    "ancsh_pose_joint_direction": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "ancsh_ransac_single": [_c_int, _vp, _vp, _vp, _c_float, _c_int, _vp, ctypes.c_ulonglong, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_ransac_single_ex": [_c_int, _vp, _vp, _vp, _c_float, _c_int, _vp, ctypes.c_ulonglong, _c_int, _vp, _vp, _vp, _vp, _vp, _c_long, _vp],
    "ancsh_ransac_single_rec": [_c_int, _vp, _vp, _vp, _c_float, _c_int, _vp, ctypes.c_ulonglong, _c_int, _vp, _vp, _vp, _vp, _vp, _c_long,
                                _vp, _c_int, _vp, _c_float, _vp],
    "ancsh_ransac_joint_rec": [_c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _c_int, _vp, ctypes.c_ulonglong, _c_int]
                              + [_vp] * 7 + [_c_int, _vp, _c_int, _vp, ctypes.c_double, _vp],
    "ancsh_ransac_joint": [_c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _c_int, _vp, ctypes.c_ulonglong, _c_int]
                          + [_vp] * 7 + [_vp],
    "ancsh_input_sample": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_test_losses": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp],
    "ancsh_ransac_joint_ex": [_c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _c_int, _vp, ctypes.c_ulonglong, _c_int]
                             + [_vp] * 7 + [_c_int, _vp],
    "ancsh_umeyama": [_c_int, _vp, _vp, _vp, _vp, _vp],
    "ancsh_estimate_similarity_transform": [_c_int, _vp, _vp, _vp, _c_int, _vp, ctypes.c_ulonglong, _vp, _vp, _vp],
}

_lib = None


def build(force=False):
    """Compile every HIP source for gfx950 into the in-tree libancsh_hip.so (hipcc cross-compiles
    without a GPU)."""
    args = ["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built (run __graft_entry__.build()); "
                "there is no CPU fallback for the product path")
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the ABI is incomplete
            fn.argtypes = argtypes
            fn.restype = _c_int
        L.ancsh_last_error.restype = ctypes.c_char_p
        L.ancsh_abi_version.restype = _c_int
        L.ancsh_last_ball_query_schedule.restype = _c_int
        L.ancsh_sa_packed_weight_floats.argtypes = [_c_int, _c_int]
        L.ancsh_sa_packed_weight_floats.restype = _c_long
        L.ancsh_sa_packed_weight_bytes_bf16x3.argtypes = [_c_int, _c_int]
        L.ancsh_sa_packed_weight_bytes_bf16x3.restype = _c_long
        L.ancsh_sa_packed_weight_bytes_f16x2.argtypes = [_c_int, _c_int]
        L.ancsh_sa_packed_weight_bytes_f16x2.restype = _c_long
        L.ancsh_ransac_single_quads_floats.argtypes = [_c_long, _c_int]
        L.ancsh_ransac_single_quads_floats.restype = _c_long
        _lib = L
    return _lib


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


_prof = None   # when a list: every call is bracketed by HIP events on the launch stream
_prof_lead = (0, {})


def profile_start(lead=0, lead_for=None):
    """lead = n: every call is issued n times back-to-back before the timed launch (all entry points are pure functions of
    their inputs, so repeating one is harmless); lead_for = {abi name: n} overrides it per entry point.  The lead launches keep the chip under load between the timed ones: a kernel
    that starts on an idle chip runs at the clock the power management is still ramping (measured with s_memtime: the fused SA
    kernels take 238 / 284 us in a 20-launch loop from idle and 209 / 262 us once the loop has run for 50 ms), and the eager pass
    spends more time in Python between launches than on the GPU."""
    global _prof, _prof_lead
    _prof, _prof_lead = [], (int(lead), dict(lead_for or {}))


def profile_stop():
    """-> [(abi name, args, milliseconds)] for every call since profile_start(); synchronises."""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    return [(n, a, e0.elapsed_time(e1)) for n, a, e0, e1 in rec]


def call(name, *args):
    """Call an ABI function on torch's current HIP stream; raise ValueError on a bad argument
    (mirrors the reference's OP_REQUIRES -> InvalidArgument) and RuntimeError on a HIP failure."""
    L = lib()
    if _prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(_prof_lead[1].get(name, _prof_lead[0])):
            getattr(L, name)(*args, stream_ptr())
        e0.record()
        rc = getattr(L, name)(*args, stream_ptr())
        e1.record()
        _prof.append((name, args, e0, e1))
    else:
        rc = getattr(L, name)(*args, stream_ptr())
    if rc != 0:
        msg = L.ancsh_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise RuntimeError(f"{name}: {msg}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("articulated-pose_amd ops run on the MI355X only: got a CPU tensor "
                               "(no CPU fallback in the product path)")
