"""EXPERIMENT (opt-in): the fused set-abstraction levels with f32 products emulated by six bf16 products on
v_mfma_f32_32x32x16_bf16 (csrc/sa_bf16x3.hip).  f32 stays the arithmetic of record; these tests pin what the experiment claims:
the level's outputs agree with the f32 kernel to f32 summation noise, the whole network stays inside the 1e-4 parity bar with the
part labels unchanged on the test clouds, and the report (max |diff|, label flips, kernel time) is written to gpurun_out/."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _level(dev, B, n, m, cf, mlp, seed, bf16):
    from articulated_pose_amd import _lib, tf_ops
    from articulated_pose_amd.tf_ops.tf_sampling import farthest_point_sample_gather
    rng = np.random.RandomState(seed)
    xyz = torch.from_numpy((rng.rand(B, n, 3).astype(np.float32) - 0.5)).to(dev)
    _, new_xyz = farthest_point_sample_gather(m, xyz)
    idx, _ = tf_ops.query_ball_point(0.25 if cf == 0 else 0.4, 64, xyz, new_xyz)
    feats = None if cf == 0 else torch.from_numpy(rng.randn(B, n, cf).astype(np.float32)).to(dev)
    cin, params, keep = 3 + cf, [], []
    for c in mlp:
        w = torch.from_numpy((rng.randn(cin, c) / np.sqrt(cin)).astype(np.float32)).to(dev)
        if bf16:
            pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_bytes_bf16x3(cin, c), dtype=torch.uint8, device=dev)
            _lib.call("ancsh_sa_pack_weights_bf16x3", cin, c, _lib.ptr(w), _lib.ptr(pk))
        else:
            pk = w
        layer = [pk, torch.from_numpy(rng.randn(c).astype(np.float32) * 0.1).to(dev), torch.from_numpy((rng.rand(c) + 0.5).astype(np.float32)).to(dev),
                 torch.from_numpy(rng.randn(c).astype(np.float32) * 0.1).to(dev)]
        params += layer
        keep.append(w)
        cin = c
    return xyz, new_xyz, idx, feats, params, keep


@pytest.mark.parametrize("cf,mlp,n,m", [(0, (64, 64, 128), 1024, 512), (128, (128, 128, 256), 512, 128)])
def test_bf16x3_level_matches_f32_chain(dev, cf, mlp, n, m):
    """a level through the bf16x3 kernels against a float64 evaluation, measured against f32's own re-association noise; the level
    with features goes through the partial-sum entry (f32 per-point partial sums of the first layer's feature part, as the f32 path)"""
    from articulated_pose_amd import _lib
    B = 4
    xyz, new_xyz, idx, feats, params, ws = _level(dev, B, n, m, cf, mlp, 7, True)
    out = torch.empty((B, m, mlp[2]), dtype=torch.float32, device=dev)
    if cf == 0:
        ptrs = (ctypes.c_void_p * 12)(*[p.data_ptr() for p in params])
        _lib.call("ancsh_sa_module_fused_bf16x3", B, n, m, 64, 0, *mlp, _lib.ptr(xyz), None, _lib.ptr(new_xyz), _lib.ptr(idx),
                  ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
    else:
        w0 = ws[0]                                                   # rows [x - c (3) | features (cf)]
        partial = torch.empty((B, n, mlp[0]), dtype=torch.float32, device=dev)
        wf = w0[3:].contiguous()
        _lib.call("ancsh_conv1x1", B * n, cf, mlp[0], _lib.ptr(feats), cf, _lib.ptr(wf), None, None, None, 2, _lib.ptr(partial), mlp[0], 0)
        wx = w0[:3].contiguous()
        pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_bytes_bf16x3(3, mlp[0]), dtype=torch.uint8, device=dev)
        _lib.call("ancsh_sa_pack_weights_bf16x3", 3, mlp[0], _lib.ptr(wx), _lib.ptr(pk))
        ptrs = (ctypes.c_void_p * 12)(*([pk.data_ptr()] + [p.data_ptr() for p in params[1:]]))
        _lib.call("ancsh_sa_module_fused_partial_bf16x3", B, n, m, 64, *mlp, _lib.ptr(xyz), _lib.ptr(partial), _lib.ptr(new_xyz), _lib.ptr(idx),
                  ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
    # float64 reference of the same level (gather, three layers, max)
    ii = idx.long()
    bi = torch.arange(B, device=dev).view(B, 1, 1)
    g = xyz.double()[bi, ii] - new_xyz.double().unsqueeze(2)
    x = g if cf == 0 else torch.cat([g, feats.double()[bi, ii]], dim=-1)
    x32 = x.float()
    for i, w in enumerate(ws):
        b, sc, sh = (params[4 * i + j] for j in (1, 2, 3))
        x = torch.relu((x @ w.double() + b.double()) * sc.double() + sh.double())
        x32 = torch.relu((x32 @ w + b) * sc + sh)                                   # an f32 evaluation with yet another order
    want, f32_alt = x.max(dim=2).values, x32.max(dim=2).values
    err = float((out.double() - want).abs().max())
    noise = float((f32_alt.double() - want).abs().max())
    scale = float(want.abs().max())
    assert err <= 4e-6 * max(1.0, scale), (err, noise, scale)
    assert err <= 8 * max(noise, 1e-7 * scale), (err, noise)                       # no worse than plain f32 re-association


def test_bf16x3_network_parity_report(dev, monkeypatch):
    """whole forwards with both SA levels on the bf16 pipe: labels equal to the f32 path's on every test cloud, floats within 1e-5
    of it (bar of the experiment) and within 1e-4 of the CPU oracle (the parity bar); the report goes to gpurun_out/"""
    from articulated_pose_amd import pointnet_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    from test_network_gpu import synth_cloud
    report = {"cases": []}
    worst, flips, points = 0.0, 0, 0
    for K, N, B, nocs_type in ((3, 1024, 8, "ancsh"), (3, 1024, 8, "npcs"), (2, 2048, 4, "ancsh"), (4, 2048, 4, "npcs")):
        mixed = nocs_type == "ancsh"
        w = synthetic_weights(K, mixed_pred=mixed, early_split_nocs=mixed, seed=K)
        P = synth_cloud(np.random.RandomState(31 * K + N), B, N)
        net = Network(K, w, nocs_type, dev)
        monkeypatch.setattr(pointnet_util, "SA_BF16X3", 0)
        f32 = {k: v.cpu().numpy() for k, v in net.predict(P).items()}
        monkeypatch.setattr(pointnet_util, "SA_BF16X3", 2)
        b16 = {k: v.cpu().numpy() for k, v in net.predict(P).items()}
        monkeypatch.setattr(pointnet_util, "SA_BF16X3", 0)
        ora = net_oracle.forward(w, P[:2], K, mixed_pred=mixed, early_split_nocs=mixed)
        d = max(float(np.abs(b16[k] - f32[k]).max()) for k in f32)
        do = max(float(np.abs(b16[k][:2] - ora[k]).max()) for k in ora)
        fl = int((b16["W"].argmax(2) != f32["W"].argmax(2)).sum())
        report["cases"].append(dict(K=K, N=N, B=B, nocs_type=nocs_type, max_abs_diff_vs_f32_path=d, max_abs_diff_vs_cpu_oracle=do, label_flips=fl))
        worst, flips, points = max(worst, d), flips + fl, points + B * N
        assert do <= 1e-4, (K, N, nocs_type, do)
    report.update(max_abs_diff_vs_f32_path=worst, label_flips=flips, points=points)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(report, open(os.path.join(out, "bf16x3_parity.json"), "w"), indent=1)
    assert flips == 0 and worst <= 1e-5, report


def test_bf16x3_level_time_report(dev):
    """kernel time of the two levels, bf16x3 against the f32 kernels (back-to-back launches, loaded clock); report only"""
    from articulated_pose_amd import _lib
    B, res = 32, {}
    for name, cf, mlp, n, m in (("SA1", 0, (64, 64, 128), 1024, 512), ("SA2_partial_sums", -1, (128, 128, 256), 512, 128)):
        xyz, new_xyz, idx, feats, params, ws = _level(dev, B, n, m, max(cf, 0), mlp, 3, True)
        out = torch.empty((B, m, mlp[2]), dtype=torch.float32, device=dev)
        ptrs = (ctypes.c_void_p * 12)(*[p.data_ptr() for p in params])
        if cf < 0:      # the register-resident kernel on per-point partial sums (any values: timing only)
            partial = torch.randn(B, n, mlp[0], device=dev)
            fn = lambda: _lib.call("ancsh_sa_module_fused_partial_bf16x3", B, n, m, 64, *mlp, _lib.ptr(xyz), _lib.ptr(partial), _lib.ptr(new_xyz),
                                   _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
        else:
            fn = lambda: _lib.call("ancsh_sa_module_fused_bf16x3", B, n, m, 64, cf, *mlp, _lib.ptr(xyz), _lib.ptr(feats), _lib.ptr(new_xyz),
                                   _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
        for _ in range(300):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + "_bf16x3_us"] = round(e0.elapsed_time(e1) / 300 * 1e3, 1)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(res, open(os.path.join(out, "bf16x3_time.json"), "w"), indent=1)
    print(res)


@pytest.mark.parametrize("K,N,B", [(3, 1024, 3), (2, 2048, 2)])
def test_bf16x3_paired_forward_equals_separate_forwards(dev, monkeypatch, K, N, B):
    """With the experiment on, the PAIRED forward (both networks per launch, what AncshPipeline runs) takes the grouped bf16x3 SA
    launches -- same kernel, same arithmetic per neighbourhood: every output equal, bit for bit, to each network's own forward."""
    from articulated_pose_amd import pointnet_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from test_network_gpu import synth_cloud
    P = synth_cloud(np.random.RandomState(5 * K + N), B, N)
    net_a = Network(K, synthetic_weights(K, seed=3), "ancsh", dev)
    net_n = Network(K, synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=4), "npcs", dev)
    monkeypatch.setattr(pointnet_util, "SA_BF16X3", 2)
    pair = PairedNetworks([net_a, net_n])
    assert pair.eligible()
    pa, pn = pair.predict(P)
    sa, sn = net_a.predict(P), net_n.predict(P)
    for got, want in ((pa, sa), (pn, sn)):
        for k in want:
            assert torch.equal(got[k], want[k]), k
    monkeypatch.setattr(pointnet_util, "SA_BF16X3", 0)
    fa = net_a.predict(P)
    assert max(float((fa[k] - sa[k]).abs().max()) for k in fa) <= 1e-5 and not torch.equal(fa["nocs_per_point"], sa["nocs_per_point"])


@pytest.mark.parametrize("K,N,B", [(3, 1024, 2), (2, 2048, 2), (4, 2048, 2)])
def test_bf16x3_whole_path_against_the_oracle(dev, monkeypatch, K, N, B):
    """Level 3 of the experiment (both SA levels + the tail chain on the bf16 pipe, csrc/tail_bf16x3.hip; the mid-section stays f32) on the
    three BASELINE shapes, both networks through the paired forward, against the CPU ORACLE (not the f32 path): integer part labels exact,
    every float head within 1e-5 -- ten times inside the north star's 1e-4."""
    from articulated_pose_amd import pointnet_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    from test_network_gpu import synth_cloud
    P = synth_cloud(np.random.RandomState(17 * K + N), B, N)
    w_a = synthetic_weights(K, seed=K)
    w_n = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=K + 1)
    pair = PairedNetworks([Network(K, w_a, "ancsh", dev), Network(K, w_n, "npcs", dev)])
    monkeypatch.setattr(pointnet_util, "SA_BF16X3", 3)
    assert pair.eligible()
    got = pair.predict(P)
    monkeypatch.setattr(pointnet_util, "SA_BF16X3", 0)
    f32 = pair.predict(P)
    report = {}
    for name, w, mixed, g, f in (("ancsh", w_a, True, got[0], f32[0]), ("npcs", w_n, False, got[1], f32[1])):
        want = net_oracle.forward(w, P, K, mixed_pred=mixed, early_split_nocs=mixed)
        gn = {k: v.cpu().numpy() for k, v in g.items()}
        assert set(gn) == set(want)
        np.testing.assert_array_equal(gn["W"].argmax(2), want["W"].argmax(2), err_msg=name)
        err = {k: float(np.abs(gn[k] - want[k]).max()) for k in want}
        assert max(err.values()) <= 1e-5, (name, err)
        assert not torch.equal(g["nocs_per_point"], f["nocs_per_point"])                # the experiment's arithmetic did run
        report[name] = max(err.values())
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "bf16x3_level3_vs_oracle_K%d_N%d.json" % (K, N)), "w") as fh:
            json.dump(report, fh)


def test_bf16x3_tail_rejects_other_program_shapes(dev):
    """The register-tile kernel is straight-line code for ONE program shape; anything else is refused before the launch."""
    from articulated_pose_amd import _lib
    L = _lib.lib()
    p16 = ctypes.c_void_p(256)
    def call(shape):                                   # shape: list of (k, n, act, has_out)
        n = len(shape)
        ops = (ctypes.c_int * (5 * n))(*[v for k, c, a, o in shape for v in (k, c, a, 0, 40 if o else 0)])
        ptrs = (ctypes.c_void_p * (5 * n))(*[256] * (5 * n))
        for i, (_k, _c, _a, o) in enumerate(shape):
            if not o:
                ptrs[5 * i + 4] = None
        nops = (ctypes.c_int * 1)(n)
        ot = (ctypes.c_void_p * 1)(ctypes.cast(ops, ctypes.c_void_p))
        pt = (ctypes.c_void_p * 1)(ctypes.cast(ptrs, ctypes.c_void_p))
        return L.ancsh_mlp_chain_grouped_fp_bf16x3(1, 0, 1024, 512, 128, p16, p16, p16, p16, ctypes.cast(nops, ctypes.c_void_p),
                                                   ctypes.cast(ot, ctypes.c_void_p), ctypes.cast(pt, ctypes.c_void_p), None)
    F, H, Lin, h = (131, 128, 1, False), (128, 128, 1, False), (128, 128, 0, False), (128, 9, 0, True)
    assert call([F, H, H, H, h, Lin, h, H, H, h]) == 0 and call([F, H, H, H, h, h, H, H, h]) == 0          # b = 0: validated, nothing launched
    assert call([F, H, H, h, H, H, h]) == -1 and b"is not F H H H" in L.ancsh_last_error()
    assert call([F, H, H, H, h, H, h, H, H, h]) == -1                                                    # the split layer must be linear
    assert call([H, H, H, H, h, H, H, h]) == -1 and b"op 0" in L.ancsh_last_error()
    assert call([F, H, H, H, (128, 33, 0, True), H, H, h]) == -1 and b"head block" in L.ancsh_last_error()


@pytest.mark.parametrize("K,N,B", [(3, 1024, 2), (2, 2048, 2), (4, 2048, 2)])
def test_f16x2_whole_path_against_the_oracle(dev, monkeypatch, K, N, B):
    """The F16x2 scheme (csrc/bx3.h: two f16 terms per operand, three products into two accumulators -- half the matrix work of bf16x3, ~22
    bits per operand) at level 3 on the three BASELINE shapes, both networks through the paired forward, against the CPU ORACLE: integer
    part labels exact, every float head within 1e-5."""
    from articulated_pose_amd import pointnet_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    from test_network_gpu import synth_cloud
    P = synth_cloud(np.random.RandomState(17 * K + N), B, N)
    w_a = synthetic_weights(K, seed=K)
    w_n = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=K + 1)
    pair = PairedNetworks([Network(K, w_a, "ancsh", dev), Network(K, w_n, "npcs", dev)])
    monkeypatch.setattr(pointnet_util, "SPLIT_SCHEME", "f16x2")
    monkeypatch.setattr(pointnet_util, "SA_BF16X3", 3)
    got = pair.predict(P)
    monkeypatch.setattr(pointnet_util, "SPLIT_SCHEME", "bf16x3")
    b16 = pair.predict(P)
    report = {}
    for name, w, mixed, g, b in (("ancsh", w_a, True, got[0], b16[0]), ("npcs", w_n, False, got[1], b16[1])):
        want = net_oracle.forward(w, P, K, mixed_pred=mixed, early_split_nocs=mixed)
        gn = {k: v.cpu().numpy() for k, v in g.items()}
        np.testing.assert_array_equal(gn["W"].argmax(2), want["W"].argmax(2), err_msg=name)
        err = {k: float(np.abs(gn[k] - want[k]).max()) for k in want}
        report[name] = dict(f16x2_vs_oracle=max(err.values()), bf16x3_vs_oracle=max(float(np.abs(b[k].cpu().numpy() - want[k]).max()) for k in want))
        assert max(err.values()) <= 1e-5, (name, err)
        assert not torch.equal(g["nocs_per_point"], b["nocs_per_point"])                # another arithmetic did run
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "f16x2_level3_vs_oracle_K%d_N%d.json" % (K, N)), "w") as fh:
            json.dump(report, fh)


@pytest.mark.parametrize("scheme", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("cf,mlp,n,m", [(0, (64, 64, 128), 1024, 512), (128, (128, 128, 256), 512, 128)])
def test_split16_level_against_float64(dev, monkeypatch, scheme, cf, mlp, n, m):
    """One set-abstraction level through each scheme's grouped entry point (two 'networks' = the same layers twice) against a float64
    evaluation: bf16x3 within f32's own re-association noise, f16x2 within 8 x 2^-22 of the output scale per layer chain."""
    from articulated_pose_amd import _lib, pointnet_util
    monkeypatch.setattr(pointnet_util, "SPLIT_SCHEME", scheme)
    B = 3
    xyz, new_xyz, idx, feats, params, ws = _level(dev, B, n, m, cf, mlp, 11, False)
    packed = [pointnet_util._split_pack(w if i or cf == 0 else w[:3].contiguous()) for i, w in enumerate(ws)]
    per_net = [x for i in range(3) for x in (packed[i], params[4 * i + 1], params[4 * i + 2], params[4 * i + 3])]
    ptrs = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in per_net + per_net])
    out = torch.empty((2 * B, m, mlp[2]), dtype=torch.float32, device=dev)
    if cf == 0:
        _lib.call(pointnet_util.split_name("ancsh_sa_module_fused_bf16x3_grouped"), 2, B, n, m, 64, 0, *mlp, _lib.ptr(xyz), None, _lib.ptr(new_xyz),
                  _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
    else:
        partial = torch.empty((B, n, mlp[0]), dtype=torch.float32, device=dev)
        wf = ws[0][3:].contiguous()
        _lib.call("ancsh_conv1x1", B * n, cf, mlp[0], _lib.ptr(feats), cf, _lib.ptr(wf), None, None, None, 2, _lib.ptr(partial), mlp[0], 0)
        partial2 = torch.cat([partial, partial], dim=0).contiguous()
        _lib.call(pointnet_util.split_name("ancsh_sa_module_fused_partial_bf16x3_grouped"), 2, B, n, m, 64, *mlp, _lib.ptr(xyz), _lib.ptr(partial2),
                  _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
    ii = idx.long()
    bi = torch.arange(B, device=dev).view(B, 1, 1)
    g = xyz.double()[bi, ii] - new_xyz.double().unsqueeze(2)
    x = g if cf == 0 else torch.cat([g, feats.double()[bi, ii]], dim=-1)
    for i, w in enumerate(ws):
        b, sc, sh = (params[4 * i + j] for j in (1, 2, 3))
        x = torch.relu((x @ w.double() + b.double()) * sc.double() + sh.double())
    want = x.max(dim=2).values
    assert torch.equal(out[:B], out[B:])                                              # both 'networks' of the grouped launch
    err, scale = float((out[:B].double() - want).abs().max()), float(want.abs().max())
    assert err <= (4e-6 if scheme == "bf16x3" else 2e-5) * max(1.0, scale), (scheme, err, scale)


@pytest.mark.parametrize("scheme", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("K,N,B", [(3, 1024, 2), (2, 2048, 2), (4, 2048, 1)])
def test_split16_level4_against_the_oracle(dev, monkeypatch, scheme, K, N, B):
    """Level 4 of the experiment: the mid-section (layer3, fa_layer1, fa_layer2: csrc/mid_bf16x3.hip, activations as 16-bit planes in LDS) on the
    16-bit pipe too -- every shared-MLP layer of the network except SA2's per-point partial conv and fa_layer1's single-source product (exact
    f32 by construction).  Both networks, paired forward, against the CPU ORACLE on the three BASELINE shapes: labels exact, floats 1e-5."""
    from articulated_pose_amd import pointnet_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    from test_network_gpu import synth_cloud
    P = synth_cloud(np.random.RandomState(23 * K + N), B, N)
    w_a = synthetic_weights(K, seed=K + 5)
    w_n = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=K + 6)
    pair = PairedNetworks([Network(K, w_a, "ancsh", dev), Network(K, w_n, "npcs", dev)])
    monkeypatch.setattr(pointnet_util, "SPLIT_SCHEME", scheme)
    monkeypatch.setattr(pointnet_util, "SA_BF16X3", 4)
    got = pair.predict(P)
    monkeypatch.setattr(pointnet_util, "SA_BF16X3", 3)
    l3 = pair.predict(P)
    report = {}
    for name, w, mixed, g, h in (("ancsh", w_a, True, got[0], l3[0]), ("npcs", w_n, False, got[1], l3[1])):
        want = net_oracle.forward(w, P, K, mixed_pred=mixed, early_split_nocs=mixed)
        gn = {k: v.cpu().numpy() for k, v in g.items()}
        np.testing.assert_array_equal(gn["W"].argmax(2), want["W"].argmax(2), err_msg=name)
        err = {k: float(np.abs(gn[k] - want[k]).max()) for k in want}
        report[name] = max(err.values())
        assert max(err.values()) <= 1e-5, (scheme, name, err)
        assert not torch.equal(g["nocs_per_point"], h["nocs_per_point"])                # the mid-section's other arithmetic did run
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "%s_level4_vs_oracle_K%d_N%d.json" % (scheme, K, N)), "w") as fh:
            json.dump(report, fh)


def test_pipeline_arithmetic_switch(dev):
    """AncshPipeline(arithmetic=...) pins the shared-MLP arithmetic per pipeline: "f16x2" gives the split-16 outputs (close to, not equal to,
    the f32 ones), the module-level defaults are left as they were, and the pose records of the two pipelines agree to the fit's tolerance."""
    from articulated_pose_amd import pointnet_util
    from articulated_pose_amd.pipeline import AncshPipeline
    from articulated_pose_amd.synthetic import passthrough_pose_problem
    K, B, N = 3, 2, 1024
    pb = passthrough_pose_problem(K, B, N, seed=3)
    before = (pointnet_util.SA_BF16X3, pointnet_util.SPLIT_SCHEME)
    outs = {}
    for arith in ("f32", "f16x2"):
        pipe = AncshPipeline(K, pb["w_ancsh"], pb["w_npcs"], B, N, dev, couple=True, use_graph=False, niter_a=500, niter_b=16, slots=1, arithmetic=arith)
        pipe.load_inputs(pb["P"], pb["cls"])
        sl, out = pipe.step()
        sl.stream.synchronize()
        outs[arith] = (out["npcs"]["nocs_per_point"].clone(), out["record"].clone())
    assert (pointnet_util.SA_BF16X3, pointnet_util.SPLIT_SCHEME) == before
    d = float((outs["f32"][0] - outs["f16x2"][0]).abs().max())
    assert 0 < d <= 1e-5, d
    assert torch.isfinite(outs["f16x2"][1]).all()
    with pytest.raises(ValueError):
        AncshPipeline(K, pb["w_ancsh"], pb["w_npcs"], B, N, dev, arithmetic="fp8")


SPLIT_SEEDS = range(int(os.environ.get("ANCSH_SPLIT16_SWEEP_SEEDS", "6")))


@pytest.mark.parametrize("seed", SPLIT_SEEDS)
def test_split16_forward_sweep(dev, monkeypatch, seed):
    """Seeded sweep of whole paired forwards at the experiment's highest level over ragged shapes -- K = 2 / 3 / 4, 512..2600 points (multiples of
    64 take the split-16 tail, the others fall back to the f32 tail; the SA levels and the mid-section are split-16 either way), 1..4 clouds, the
    two schemes alternating -- against the CPU oracle: labels exact, floats 1e-5."""
    from articulated_pose_amd import pointnet_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    from test_network_gpu import synth_cloud
    rng = np.random.RandomState(7000 + seed)
    K = int(rng.choice([2, 3, 4]))
    N = int(rng.choice([512, 640, 1024, 1536, 2048, 2560])) if seed % 3 else int(rng.randint(512, 2600))
    B = int(rng.randint(1, 5))
    scheme = ("f16x2", "bf16x3")[seed % 2]
    P = synth_cloud(rng, B, N)
    w_a = synthetic_weights(K, seed=30 + seed)
    w_n = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=60 + seed)
    pair = PairedNetworks([Network(K, w_a, "ancsh", dev), Network(K, w_n, "npcs", dev)])
    monkeypatch.setattr(pointnet_util, "SPLIT_SCHEME", scheme)
    monkeypatch.setattr(pointnet_util, "SA_BF16X3", 4)
    got = pair.predict(P)
    for name, w, mixed, g in (("ancsh", w_a, True, got[0]), ("npcs", w_n, False, got[1])):
        want = net_oracle.forward(w, P, K, mixed_pred=mixed, early_split_nocs=mixed)
        gn = {k: v.cpu().numpy() for k, v in g.items()}
        err = max(float(np.abs(gn[k] - want[k]).max()) for k in want)
        assert err <= 1e-5, (scheme, K, N, B, name, err)
        # part labels: equal wherever the oracle's two largest probabilities are further apart than the arithmetic's summation noise.  The
        # split-16 sums are ordered by the MFMA, not by k, so an argmax NEAR-TIE can fall the other way: in the 300-seed sweep of round 6
        # (2.2 M points) exactly one label differed -- seed 253, its top-2 margin printed below -- never a label with a real margin.
        flipped = np.argwhere(gn["W"].argmax(2) != want["W"].argmax(2))
        top2 = np.sort(want["W"], axis=2)[:, :, -2:]
        for b, i in flipped:
            margin = float(top2[b, i, 1] - top2[b, i, 0])
            print("near-tie label: %s K=%d N=%d B=%d %s cloud %d point %d margin %.3e" % (scheme, K, N, B, name, b, i, margin))
            assert margin <= 2e-6, (scheme, K, N, B, name, b, i, margin)
        assert len(flipped) <= 1, (scheme, K, N, B, name, len(flipped))
