"""CPU tests (no GPU): libancsh_hip.so loads and exports every symbol include/ancsh_hip.h declares
(no compute calls), argument validation happens before any launch, and host logic round-trips."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ancsh_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ancsh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from articulated_pose_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/ancsh_hip.h but not exported"
    assert set(_lib.SIGNATURES) | {"ancsh_abi_version", "ancsh_last_error", "ancsh_sa_packed_weight_floats",
                                   "ancsh_ransac_single_quads_floats", "ancsh_sa_packed_weight_bytes_bf16x3", "ancsh_sa_packed_weight_bytes_f16x2",
                                   "ancsh_last_ball_query_schedule"} == set(syms)
    assert _lib.lib().ancsh_abi_version() >= 1


def test_measurement_aids_live_outside_the_product_library():
    """bench.py's HBM-copy yardstick has its own .so (tools/microbench): the product library exports operators only."""
    import ctypes
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from articulated_pose_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    assert not hasattr(L, "ancsh_hbm_copy") and not hasattr(L, "yardstick_hbm_copy")
    if not os.path.exists(os.path.join(bench.YARDSTICK_DIR, "libyardstick.so")):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", bench.YARDSTICK_DIR])
    Y = bench.yardstick()
    p8 = ctypes.c_void_p(8)
    assert Y.yardstick_hbm_copy(15, p8, p8, None) == -1 and Y.yardstick_hbm_copy(0, None, None, None) == 0      # argument checks come before any launch


def test_bad_arguments_are_rejected_before_launch():
    """EINVAL paths return before touching the device, so they are testable without a GPU."""
    import ctypes
    from articulated_pose_amd import _lib
    L = _lib.lib()
    assert L.ancsh_query_ball_point(1, 16, 4, -1.0, 8, None, None, None, None, None) == -1
    assert b"positive radius" in L.ancsh_last_error()
    assert L.ancsh_query_ball_point(1, 16, 4, 0.5, 0, None, None, None, None, None) == -1
    assert b"positive nsample" in L.ancsh_last_error()
    assert L.ancsh_farthest_point_sample(1, 16, 0, None, None, None, None) == -1
    assert b"positive npoint" in L.ancsh_last_error()
    assert L.ancsh_conv1x1(128, 8, 8, None, 4, None, None, None, None, 0, None, 8, 0, None) == -1
    assert L.ancsh_conv1x1(100, 8, 8, None, 8, None, None, None, None, 0, None, 8, 64, None) == -1
    assert L.ancsh_head_activations(10, 9, 1, None, 100, *([None] * 10), None) == -1
    # round-2 entry points: argument errors before any launch
    assert L.ancsh_prob_sample(1, 0, 4, None, None, None, None, None) == -1 and b"ProbSample" in L.ancsh_last_error()
    assert L.ancsh_selection_sort(1, 16, 4, 0, None, None, None, None) == -1 and b"positive k" in L.ancsh_last_error()
    assert L.ancsh_selection_sort(1, 9000, 4, 3, None, None, None, None) == -1
    assert L.ancsh_knn_point(1, 8, 4, 3, 9, None, None, None, None, None) == -1 and b"exceeds" in L.ancsh_last_error()
    assert L.ancsh_input_sample(2, 16, 2, None, None, None, None, 3, -1, 3, None, None, None, None, None) == -1
    assert L.ancsh_input_sample(2, 16, 8, None, None, None, None, 9, -1, 3, None, None, None, None, None) == -1
    assert L.ancsh_test_losses(2, 16, 9, 0, None, None, None) == -1 and b"n_max_parts" in L.ancsh_last_error()
    assert L.ancsh_test_losses(2, 16, 3, 2, None, None, None) == -1 and b"Soft_L1" in L.ancsh_last_error()
    assert L.ancsh_query_ball_point_multi(5, None, None, None, None, None, None, None, None, None, None) == -1
    assert L.ancsh_group_point_multi(0, None, None, None, None, None, None, None, None, None) == -1
    assert L.ancsh_farthest_point_sample(1, 9000, 4, ctypes.c_void_p(8), None, ctypes.c_void_p(8), None) == -1     # n > 8192 needs temp
    assert b"temp" in L.ancsh_last_error()
    p8 = ctypes.c_void_p(8)
    assert L.ancsh_sa_module_fused_partial(1, 16, 4, 32, 128, 128, 256, p8, p8, p8, p8, p8, p8, None) == -1 and b"nsample must be 64" in L.ancsh_last_error()
    assert L.ancsh_sa_module_fused_partial(1, 16, 4, 64, 128, 128, 256, p8, None, p8, p8, p8, p8, None) == -1 and b"null pointer" in L.ancsh_last_error()
    assert L.ancsh_sa_module_fused_partial(1, 16, 4, 64, 128, 128, 256, p8, p8, p8, p8, p8, p8, None) == -1 and b"16-byte aligned" in L.ancsh_last_error()
    assert L.ancsh_sa_module_fused_partial(0, 16, 4, 64, 128, 128, 256, None, None, None, None, None, None, None) == 0
    assert L.ancsh_ransac_joint_ex(1, *([None] * 5), 0.1, 8, None, 0, 16, *([None] * 7), 7, None) == -1 and b"lm_schedule" in L.ancsh_last_error()
    # round-3 entry points
    assert L.ancsh_ransac_single_ex(1, p8, p8, p8, 0.1, 8, None, 0, 16, p8, p8, p8, p8, None, 64, None) == -1 and b"scratch_quads is NULL" in L.ancsh_last_error()
    assert L.ancsh_ransac_single_ex(1, p8, p8, p8, 0.1, 8, None, 0, 16, p8, p8, p8, p8, p8, 64, None) == -1 and b"32-byte aligned" in L.ancsh_last_error()
    assert L.ancsh_ransac_single_ex(1, p8, p8, p8, 0.1, 8, None, 0, 16, p8, p8, p8, p8, ctypes.c_void_p(64), -1, None) == -1 and b"out of range" in L.ancsh_last_error()
    assert L.ancsh_ransac_single_ex(0, None, None, None, 0.1, 8, None, 0, 16, None, None, None, None, ctypes.c_void_p(64), 0, None) == 0
    # round-4 entry points (the one-launch ancsh_group_point_multi has a 1-D grid: no 65535-cloud limit any more)
    assert L.ancsh_query_ball_group_xyz_multi(5, *([None] * 12), None) == -1 and b"nprob" in L.ancsh_last_error()
    assert L.ancsh_mlp_chain_grouped(3, 64, 131, p8, 132, p8, p8, p8, None, None) == -1 and b"ngroups" in L.ancsh_last_error()
    assert L.ancsh_mlp_chain_grouped(1, 64, 131, p8, 132, p8, p8, p8, p8, None) == -1 and b"16-byte aligned" in L.ancsh_last_error()
    assert L.ancsh_joint_params(1, 16, 3, 5, 0, p8, None, None, p8, p8, p8, p8, None, p8, None) == -1 and b"channels" in L.ancsh_last_error()
    assert L.ancsh_joint_params(1, 16, 3, 9, 0, p8, None, None, p8, p8, p8, p8, None, p8, None) == -1 and b"needs the part mask" in L.ancsh_last_error()
    assert L.ancsh_part_extents(1, 16, 9, 27, p8, p8, p8, 3, p8, p8, p8, p8, None) == -1 and b"bad sizes" in L.ancsh_last_error()      # K <= 8
    assert L.ancsh_part_extents(1, 16, 3, 5, p8, p8, p8, 3, p8, p8, p8, p8, None) == -1 and b"channels" in L.ancsh_last_error()
    assert L.ancsh_part_extents(1, 16, 3, 9, p8, None, p8, 3, p8, p8, p8, p8, None) == -1 and b"null pointer" in L.ancsh_last_error()
    # round-5 entry points: the mid-section chains serve the ANCSH backbone widths only
    p16 = ctypes.c_void_p(16)
    assert L.ancsh_sa3_chain_grouped(5, 1, 128, 256, 256, 512, 1024, p16, p16, p16, p16, None) == -1 and b"ngroups" in L.ancsh_last_error()
    assert L.ancsh_sa3_chain_grouped(2, 1, 100, 256, 256, 512, 1024, p16, p16, p16, p16, None) == -1 and b"multiple of 32" in L.ancsh_last_error()
    assert L.ancsh_sa3_chain_grouped(2, 1, 128, 256, 256, 500, 1024, p16, p16, p16, p16, None) == -1 and b"unsupported layer shape" in L.ancsh_last_error()
    assert L.ancsh_sa3_chain_grouped(2, 1, 128, 256, 256, 512, 1024, p16, p8, p16, p16, None) == -1 and b"16-byte aligned" in L.ancsh_last_error()
    assert L.ancsh_sa3_chain_grouped(2, 0, 128, 256, 256, 512, 1024, None, None, None, None, None) == 0
    assert L.ancsh_fp_single_source_init(2, 4, 1000, 256, 4, p16, p16, p16, None) == -1 and b"bad shape" in L.ancsh_last_error()
    assert L.ancsh_fp_single_source_init(2, 4, 1024, 200, 4, p16, p16, p16, None) == -1 and b"bad shape" in L.ancsh_last_error()
    assert L.ancsh_fp_single_source_init(2, 4, 1024, 256, 4, p16, None, p16, None) == -1 and b"null pointer" in L.ancsh_last_error()
    assert L.ancsh_fp1_chain_grouped(2, 4, 128, 256, 128, 256, p16, p16, p16, p16, None) == -1 and b"unsupported layer shape" in L.ancsh_last_error()
    assert L.ancsh_fp1_chain_grouped(2, 4, 130, 256, 256, 256, p16, p16, p16, p16, None) == -1 and b"multiple of 32" in L.ancsh_last_error()
    assert L.ancsh_fp2_chain_grouped(2, 4, 128, 512, 256, 64, 256, 128, p16, p16, p16, p16, p16, p16, None) == -1 and b"unsupported layer shape" in L.ancsh_last_error()
    assert L.ancsh_fp2_chain_grouped(2, 4, 128, 500, 256, 128, 256, 128, p16, p16, p16, p16, p16, p16, None) == -1 and b"multiple of 32" in L.ancsh_last_error()
    assert L.ancsh_fp2_chain_grouped(2, 4, 128, 512, 256, 128, 256, 128, p16, None, p16, p16, p16, p16, None) == -1 and b"null pointer" in L.ancsh_last_error()
    # the tail chain with fa_layer3's interpolation in its tile load: 128-channel source rows, a multiple of 128 points per cloud
    assert L.ancsh_mlp_chain_grouped_fp(2, 4, 1000, 512, 128, p16, p16, p16, p16, None, None, None, None, None) == -1 and b"multiple of 128" in L.ancsh_last_error()
    assert L.ancsh_mlp_chain_grouped_fp(2, 4, 1024, 512, 64, p16, p16, p16, p16, None, None, None, None, None) == -1 and b"128 channels" in L.ancsh_last_error()
    assert L.ancsh_mlp_chain_grouped_fp(2, 4, 1024, 512, 128, p16, None, p16, p16, p16, p16, p16, None, None) == -1 and b"null pointer" in L.ancsh_last_error()
    assert L.ancsh_mlp_chain_grouped_fp(2, 4, 1024, 512, 128, p8, p16, p16, p16, p16, p16, p16, None, None) == -1 and b"16-byte aligned" in L.ancsh_last_error()
    assert L.ancsh_pose_poison_records(4, 1024, 3, p16, p16, None, None, p16, None) == -1 and b"null pointer" in L.ancsh_last_error()
    assert L.ancsh_pose_poison_records(4, 1024, 17, p16, p16, p16, None, p16, None) == -1 and b"bad shape" in L.ancsh_last_error()
    assert L.ancsh_pose_poison_records(0, 1024, 3, None, None, None, None, None, None) == 0
    # an EMPTY batch is still checked for ngroups and the program tables before it returns OK
    assert L.ancsh_mlp_chain_grouped_fp(2, 0, 1024, 512, 128, None, None, None, None, None, None, None, None, None) == -1 and b"program table" in L.ancsh_last_error()
    assert L.ancsh_mlp_chain_grouped_fp(99, 0, 1024, 512, 128, None, None, None, None, p16, p16, p16, None, None) == -1 and b"ngroups" in L.ancsh_last_error()
    assert L.ancsh_mlp_chain_grouped_fp(2, 0, 1024, 512, 128, None, None, None, None, p16, p16, p16, None, None) == 0
    # the single-source init stages its input row in dynamic LDS: nothing above the 48 KB a launch gets by default is accepted
    assert L.ancsh_fp_single_source_init(2, 4, 12288 + 128, 256, 4, p16, p16, p16, None) == -1 and b"max 12288" in L.ancsh_last_error()
    # the fits that write the pose record / the tie counts themselves: the record's geometry and the window are checked before any launch
    rec = ctypes.c_void_p(64)
    assert L.ancsh_ransac_single_rec(4, p8, p8, p8, 0.1, 8, None, 0, 16, p8, p8, p8, p8, None, 0, rec, 3, None, 0.0, None) == -1 and b"record needs" in L.ancsh_last_error()
    assert L.ancsh_ransac_single_rec(3, p8, p8, p8, 0.1, 8, None, 0, 16, p8, p8, p8, p8, None, 0, None, 3, p8, 0.2, None) == -1 and b"tie_window" in L.ancsh_last_error()
    assert L.ancsh_ransac_joint_rec(3, *([p8] * 5), 0.1, 8, None, 0, 16, *([p8] * 6), None, 0, rec, 3, None, 0.0, None) == -1 and b"record needs" in L.ancsh_last_error()
    assert L.ancsh_ransac_joint_rec(2, *([p8] * 5), 0.1, 8, None, 0, 16, *([p8] * 6), None, 0, None, 3, p8, -1.0, None) == -1 and b"tie_window" in L.ancsh_last_error()
    assert L.ancsh_ransac_single_rec(0, None, None, None, 0.1, 8, None, 0, 16, None, None, None, None, None, 0, None, 3, None, 0.0, None) == 0
    assert L.ancsh_part_extents(0, 16, 3, 9, None, None, None, 3, None, None, None, None, None) == 0
    # empty problems are no-ops
    assert L.ancsh_group_point(0, 16, 3, 4, 8, None, None, None, None) == 0
    assert L.ancsh_prob_sample(0, 4, 4, None, None, None, None, None) == 0
    assert L.ancsh_knn_point(0, 8, 4, 3, 2, None, None, None, None, None) == 0


def test_product_refuses_cpu_tensors():
    import torch
    from articulated_pose_amd import tf_ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tf_ops.farthest_point_sample(4, torch.zeros(1, 8, 3))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "articulated-pose_amd")
    for dp, _dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert "ancsh_oracle" not in src and "oracle._" not in src and "oracle/_" not in src, os.path.join(dp, f)


def test_weight_table_and_fold():
    from articulated_pose_amd.weights import fold_layer, layer_table, synthetic_weights, save_npz, load_npz
    from oracle import net_oracle
    w = synthetic_weights(3)
    names = [t[0] for t in layer_table(3)]
    assert "SPFN/est_net/layer1/conv0" in names and "SPFN/nocs_net/fc11_1" in names and "SPFN/joint_net/fc4_3" in names
    assert w["SPFN/est_net/layer2/conv0/weights"].shape == (1, 1, 131, 128)
    assert w["SPFN/est_net/fc1/weights"].shape == (1, 128, 128)
    assert w["SPFN/nocs_net/fc2_1/weights"].shape == (1, 128, 9)
    assert "SPFN/nocs_net/fc2_1/bn/gamma" not in w
    for n in names:                                   # product fold == oracle fold, bit for bit
        a, b = fold_layer(w, n), net_oracle.fold(w, n)
        for k in a:
            np.testing.assert_array_equal(a[k], b[k])
    w2 = synthetic_weights(2, mixed_pred=False, early_split_nocs=False)
    assert "SPFN/nocs_net/fc2_2/weights" in w2 and w2["SPFN/nocs_net/fc2_2/weights"].shape[-1] == 1
    assert "SPFN/nocs_net/fc11_1/weights" not in w2
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        save_npz(os.path.join(d, "w.npz"), w)
        w3 = load_npz(os.path.join(d, "w.npz"))
    assert set(w3) == set(w) and all(np.array_equal(w3[k], w[k]) for k in w)


def test_prediction_io_roundtrip(tmp_path):
    from articulated_pose_amd import prediction_io
    rng = np.random.RandomState(0)
    B, N, K = 2, 16, 3
    pred = {"W": rng.rand(B, N, K).astype(np.float32), "nocs_per_point": rng.rand(B, N, 3 * K).astype(np.float32),
            "confi_per_point": rng.rand(B, N, 1).astype(np.float32), "heatmap_per_point": rng.rand(B, N, 1).astype(np.float32),
            "unitvec_per_point": rng.rand(B, N, 3).astype(np.float32), "joint_axis_per_point": rng.rand(B, N, 3).astype(np.float32),
            "index_per_point": rng.rand(B, N, 3).astype(np.float32), "gocs_per_point": rng.rand(B, N, 3 * K).astype(np.float32)}
    batch = {"P": rng.rand(B, N, 3).astype(np.float32), "cls_gt": rng.randint(0, K, (B, N)),
             "joint_cls_gt": rng.randint(0, K, (B, N))}
    prediction_io.save_batch_nn("SPFN", pred, batch, ["a_0_0", "b_0_1"], str(tmp_path), is_mixed=True, W_reduced=False)
    rec = prediction_io.load_record(str(tmp_path), "b_0_1")
    np.testing.assert_array_equal(rec["instance_per_point"], pred["W"][1])       # W_reduced=False: not arg-maxed
    np.testing.assert_array_equal(rec["nocs_per_point"], pred["nocs_per_point"][1])
    np.testing.assert_array_equal(rec["joint_cls_gt"], batch["joint_cls_gt"][1])
    assert "gocs_per_point" in rec and "confidence_per_point" in rec


def test_quad_scratch_size_is_host_only_arithmetic():
    """ancsh_ransac_single_quads_floats: part p starts at quad 2 * (ceil(off[p] / 8) + p) and owns an even number of quads, so
    2 * (ceil(rows / 8) + nprob) + 2 records of 24 floats hold any partition of `rows` rows into `nprob` parts."""
    from articulated_pose_amd import _lib
    L = _lib.lib()
    assert L.ancsh_ransac_single_quads_floats(32768, 96) == 24 * (2 * (4096 + 96) + 2)
    assert L.ancsh_ransac_single_quads_floats(0, 0) == 48 and L.ancsh_ransac_single_quads_floats(-1, 3) == -1
    rng = np.random.RandomState(0)
    for _ in range(200):                                   # the layout rule never overlaps parts nor leaves the buffer
        nprob = int(rng.randint(1, 12))
        sizes = rng.randint(0, 40, nprob)
        off = np.concatenate([[0], np.cumsum(sizes)])
        cap = L.ancsh_ransac_single_quads_floats(int(off[-1]), nprob) // 24
        end = 0
        for p in range(nprob):
            q0 = 2 * ((off[p] + 7) // 8 + p)
            assert q0 >= end
            end = q0 + 2 * ((sizes[p] + 7) // 8)
        assert end <= cap


def test_packed_weight_size_is_host_only_arithmetic():
    """ancsh_sa_packed_weight_floats needs no device: ceil(k/2/4) slots x ceil(n/32) tiles x 64 lanes x 4 floats."""
    from articulated_pose_amd import _lib
    L = _lib.lib()
    assert L.ancsh_sa_packed_weight_floats(131, 128) == 17 * 4 * 256
    assert L.ancsh_sa_packed_weight_floats(3, 64) == 1 * 2 * 256
    assert L.ancsh_sa_packed_weight_floats(128, 9) == 16 * 1 * 256
    assert L.ancsh_sa_packed_weight_floats(0, 64) == -1 and L.ancsh_sa_packed_weight_floats(8, 0) == -1
