"""End-to-end parity of the network half on the MI355X against the CPU oracle (net_oracle.forward):
integer part labels argmax(W) bit-exact; every float head within 1e-4 (BASELINE.json tolerance)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: NOCS coords / head floats within 1e-4


def synth_cloud(rng, b, n):
    # unit-diagonal-ish clouds like the reference loader produces (pts * norm_factor, lib/dataset.py:351)
    c = rng.uniform(-0.3, 0.3, (b, 1, 3))
    return (c + rng.uniform(-0.45, 0.45, (b, n, 3)) * rng.uniform(0.3, 1.0, (b, 1, 3))).astype(np.float32)


@pytest.mark.parametrize("K,N,nocs_type,B", [(3, 1024, "ancsh", 3), (3, 1024, "npcs", 2), (2, 2048, "ancsh", 2),
                                             (4, 2048, "ancsh", 2), (4, 2048, "npcs", 1), (3, 1000, "ancsh", 1), (2, 777, "npcs", 3)])
def test_forward_matches_oracle(dev, K, N, nocs_type, B):
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    mixed = nocs_type == "ancsh"
    w = synthetic_weights(K, mixed_pred=mixed, early_split_nocs=mixed, seed=K)
    rng = np.random.RandomState(K * 100 + N)
    P = synth_cloud(rng, B, N)
    want = net_oracle.forward(w, P, K, mixed_pred=mixed, early_split_nocs=mixed)
    got = {k: v.cpu().numpy() for k, v in Network(K, w, nocs_type, dev).predict(P).items()}
    assert set(got) == set(want)
    np.testing.assert_array_equal(got["W"].argmax(2), want["W"].argmax(2))          # integer part labels
    for k in want:
        assert got[k].shape == want[k].shape, k
        err = np.abs(got[k] - want[k]).max()
        assert err <= TOL, (k, err)


# 8 seeds in the suite; ANCSH_NET_SWEEP_SEEDS=N for a one-off long fuzz (profiles/r05_ops_fuzz.txt)
import os
NET_SEEDS = range(int(os.environ.get("ANCSH_NET_SWEEP_SEEDS", "8")))


@pytest.mark.parametrize("seed", NET_SEEDS)
def test_forward_sweep(dev, seed):
    """Seeded sweep of whole forwards over ragged shapes -- K = 2 / 3 / 4, 512..2600 points (multiples of 128 and not: the tail
    programs take the interpolation-in-the-load path only for the former), 1..5 clouds, ANCSH / NPCS heads -- through all three entry
    points: the layer-API forward against the CPU oracle (labels exact, floats 1e-4), and predict_grouped and the PAIRED forward (both
    networks per launch, what the pipeline runs) against the layer-API forward bit for bit."""
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    rng = np.random.RandomState(4000 + seed)
    K = int(rng.choice([2, 3, 4]))
    N = int(rng.choice([512, 640, 1024, 1536, 2048, 2560])) if seed % 2 else int(rng.randint(512, 2600))
    B = int(rng.randint(1, 6))
    P = synth_cloud(rng, B, N)
    w_a = synthetic_weights(K, mixed_pred=True, early_split_nocs=True, seed=10 + seed)
    w_n = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=20 + seed)
    net_a, net_n = Network(K, w_a, "ancsh", dev), Network(K, w_n, "npcs", dev)
    outs = {}
    for name, net, w, mixed in (("ancsh", net_a, w_a, True), ("npcs", net_n, w_n, False)):
        want = net_oracle.forward(w, P, K, mixed_pred=mixed, early_split_nocs=mixed)
        got = net.predict(P)
        outs[name] = got
        g = {k: v.cpu().numpy() for k, v in got.items()}
        assert set(g) == set(want)
        np.testing.assert_array_equal(g["W"].argmax(2), want["W"].argmax(2), err_msg="K=%d N=%d B=%d %s" % (K, N, B, name))
        for k in want:
            err = np.abs(g[k] - want[k]).max()
            assert g[k].shape == want[k].shape and err <= TOL, (K, N, B, name, k, err)
        grouped = net.predict_grouped(torch.from_numpy(P).to(dev))
        for k in got:
            assert torch.equal(grouped[k], got[k]), (K, N, B, name, k)
    pair = PairedNetworks([net_a, net_n])
    pa, pn = pair.predict(P)
    for name, po in (("ancsh", pa), ("npcs", pn)):
        for k in outs[name]:
            assert torch.equal(po[k], outs[name][k]), (K, N, B, "paired " + name, k, pair.eligible())


@pytest.mark.parametrize("which", ["pooled", "all", "zero"])
def test_forward_with_negative_and_zero_bn_scales(dev, which):
    """Every synthetic weight set has positive BN gammas, so nothing else in the suite sends a negative or a zero folded scale through
    the epilogues (sign of the fma's multiplier, the pooled layers' implicit ReLU at 0): a network with NEGATIVE gammas in its pooled
    layers (every third column), in every layer, and with ZERO gammas must give the oracle's bits too."""
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    K, N, B = 3, 1024, 2
    w = dict(synthetic_weights(K, seed=31))
    for name in list(w):
        if not name.endswith("/bn/gamma"):
            continue
        pooled = any(name.endswith("layer%d/conv2/bn/gamma" % l) for l in (1, 2, 3))
        g = np.array(w[name], np.float32)
        if which == "zero":
            if pooled:
                g[1::4] = 0.0
        elif pooled or which == "all":
            g[::3] = -g[::3]
        w[name] = g
    P = synth_cloud(np.random.RandomState(77), B, N)
    want = net_oracle.forward(w, P, K)
    net = Network(K, w, "ancsh", dev)
    got = net.predict(P)
    g = {k: v.cpu().numpy() for k, v in got.items()}
    np.testing.assert_array_equal(g["W"].argmax(2), want["W"].argmax(2))
    for k in want:
        assert np.abs(g[k] - want[k]).max() <= TOL, (which, k)
    pa, pb = PairedNetworks([net, net]).predict(P)
    for k in got:
        assert torch.equal(pa[k], got[k]) and torch.equal(pb[k], got[k]), (which, k)
    # the SA levels themselves, bit for bit (the fused kernels' pooled epilogues), and the layer-by-layer forward (conv_packed's pooled path)
    from articulated_pose_amd import pointnet_util, tf_util
    aux = net_oracle.forward(w, P, K, return_aux=True)["_aux"]
    tf_util.set_variables(w)
    Pt = torch.from_numpy(P).to(dev)
    with tf_util.variable_scope("SPFN"), tf_util.variable_scope("est_net"):
        l1_xyz, l1_points, _ = pointnet_util.pointnet_sa_module(Pt, Pt[:, :, 3:3], 512, 0.2, 64, [64, 64, 128], None, False, False, None, "layer1")
        l2_xyz, l2_points, _ = pointnet_util.pointnet_sa_module(l1_xyz, l1_points, 128, 0.4, 64, [128, 128, 256], None, False, False, None, "layer2")
        _, l3_points, _ = pointnet_util.pointnet_sa_module(l2_xyz, l2_points, None, None, None, [256, 512, 1024], None, True, False, None, "layer3")
    np.testing.assert_array_equal(l1_points.cpu().numpy(), aux["l1_points"])
    np.testing.assert_array_equal(l2_points.cpu().numpy(), aux["l2_points"])
    np.testing.assert_array_equal(l3_points.cpu().numpy().reshape(aux["l3_points"].shape), aux["l3_points"])
    try:
        pointnet_util.FUSED_SA = False
        plain = net.predict(P)
    finally:
        pointnet_util.FUSED_SA = True
    for k in got:
        assert torch.equal(plain[k], got[k]), (which, "unfused", k)


def test_engine_graph_replay_is_deterministic(dev):
    from articulated_pose_amd.network import Network, AncshEngine
    from articulated_pose_amd.weights import synthetic_weights
    w = synthetic_weights(3)
    net = Network(3, w, "ancsh", dev)
    P = torch.from_numpy(synth_cloud(np.random.RandomState(0), 4, 1024)).to(dev)
    eager = {k: v.clone() for k, v in net.predict(P).items()}
    eng = AncshEngine(net, 4, 1024)
    for _ in range(3):
        out = eng(P)
        torch.cuda.synchronize()
        for k in eager:
            assert torch.equal(out[k], eager[k]), k
    grouped = net.predict_grouped(P)                  # what predict_and_save and the engine run: same bits as the layer-API forward
    for k in eager:
        assert torch.equal(grouped[k], eager[k]), k


def test_shared_geometry_is_bit_identical(dev):
    """The pipeline computes FPS / ball query / 3-NN once per batch and shares them between the ANCSH and NPCS
    networks: outputs must be identical to running each network on its own."""
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.pointnet_util import Geometry
    from articulated_pose_amd.weights import synthetic_weights
    P = torch.from_numpy(synth_cloud(np.random.RandomState(3), 3, 1024)).to(dev)
    a = Network(3, synthetic_weights(3, seed=0), "ancsh", dev)
    n = Network(3, synthetic_weights(3, mixed_pred=False, early_split_nocs=False, seed=1), "npcs", dev)
    alone_a, alone_n = a.predict(P), n.predict(P)
    g = Geometry()
    shared_a = a.predict(P, g)
    assert len(g) == 4                      # 2 SA levels + 2 FP levels (fa_layer1 has a single source point: no 3-NN)
    shared_n = n.predict(P, g)
    for k in alone_a:
        assert torch.equal(alone_a[k], shared_a[k]), k
    for k in alone_n:
        assert torch.equal(alone_n[k], shared_n[k]), k


@pytest.mark.parametrize("N", [1024, 2048])
def test_fused_sa_equals_unfused_bitwise(dev, N):
    """csrc/sa_fused.hip (gather -> 3 MLP layers -> max, activations in LDS) vs the op-by-op kernels and vs the
    CPU oracle's SA outputs: identical bits (same k-ordered fmaf chains)."""
    from articulated_pose_amd import pointnet_util, tf_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    w = synthetic_weights(3, seed=5)
    P = synth_cloud(np.random.RandomState(N), 2, N)
    net = Network(3, w, "ancsh", dev)
    try:
        pointnet_util.FUSED_SA = True
        fused = {k: v.clone() for k, v in net.predict(P).items()}
        pointnet_util.FUSED_SA = False
        plain = {k: v.clone() for k, v in net.predict(P).items()}
    finally:
        pointnet_util.FUSED_SA = True
    for k in plain:
        assert torch.equal(fused[k], plain[k]), k
    # SA outputs themselves against the oracle (bit-exact)
    want = net_oracle.forward(w, P, 3, return_aux=True)["_aux"]
    tf_util.set_variables(w)
    Pt = torch.from_numpy(P).to(dev)
    with tf_util.variable_scope("SPFN"), tf_util.variable_scope("est_net"):
        l1_xyz, l1_points, _ = pointnet_util.pointnet_sa_module(Pt, Pt[:, :, 3:3], 512, 0.2, 64, [64, 64, 128], None, False, False, None, "layer1")
        l2_xyz, l2_points, _ = pointnet_util.pointnet_sa_module(l1_xyz, l1_points, 128, 0.4, 64, [128, 128, 256], None, False, False, None, "layer2")
    np.testing.assert_array_equal(l1_points.cpu().numpy(), want["l1_points"])
    np.testing.assert_array_equal(l2_points.cpu().numpy(), want["l2_points"])


def test_fp_single_source_shortcut_equals_materialised_path_bitwise(dev):
    """fa_layer1 interpolates from ONE point per cloud: the per-cloud partial dot product + ancsh_conv1x1_ex(acc_init)
    must give the same bits as three_nn -> three_interpolate -> concat -> conv over all 1280 channels."""
    from articulated_pose_amd import pointnet_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    w = synthetic_weights(3, seed=21)
    P = synth_cloud(np.random.RandomState(3), 3, 1024)
    net = Network(3, w, "ancsh", dev)
    try:
        pointnet_util.FP_SINGLE_SOURCE = True
        short = {k: v.clone() for k, v in net.predict(P).items()}
        pointnet_util.FP_SINGLE_SOURCE = False
        plain = {k: v.clone() for k, v in net.predict(P).items()}
    finally:
        pointnet_util.FP_SINGLE_SOURCE = True
    for k in plain:
        assert torch.equal(short[k], plain[k]), k


def test_conv1x1_ex_chain_continuation_bitwise(dev):
    """RAW partial over the leading channels + acc_init over the rest == one call over all channels (odd sizes too)."""
    import ctypes
    from articulated_pose_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(0)
    for rows, grp, c2, c1, cout in ((384, 128, 1024, 256, 256), (200, 50, 37, 19, 70), (64, 64, 5, 3, 33)):
        nb = rows // grp
        gv = torch.randn(nb, c2, generator=g).to(dev); p1 = torch.randn(rows, c1, generator=g).to(dev)
        W = (torch.randn(c2 + c1, cout, generator=g) / 8).to(dev)
        bias, scale, shift = [torch.randn(cout, generator=g).to(dev) for _ in range(3)]
        full = torch.cat([gv.repeat_interleave(grp, 0), p1], 1).contiguous()
        want = torch.empty(rows, cout, device=dev)
        _lib.call("ancsh_conv1x1", rows, c2 + c1, cout, _lib.ptr(full), c2 + c1, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(want), cout, 0)
        init = torch.empty(nb, cout, device=dev)
        _lib.call("ancsh_conv1x1", nb, c2, cout, _lib.ptr(gv), c2, _lib.ptr(W), None, None, None, 2, _lib.ptr(init), cout, 0)
        got = torch.empty(rows, cout, device=dev)
        _lib.call("ancsh_conv1x1_ex", rows, c1, cout, _lib.ptr(p1), c1, _lib.ptr(W[c2:]), _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(got), cout, 0, _lib.ptr(init), grp)
        assert torch.equal(got, want), (rows, grp, c2, c1, cout)
    with pytest.raises(ValueError):
        _lib.call("ancsh_conv1x1_ex", 8, 4, 4, _lib.ptr(p1), 4, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(got), 4, 0, _lib.ptr(init), 0)


@pytest.mark.parametrize("rows,cin,ldx,cout,pool,act", [
    (4096, 259, 260, 256, 0, 1), (4096, 512, 512, 1024, 128, 1), (16384, 384, 384, 256, 0, 1), (16384, 256, 256, 128, 0, 1),
    (1000, 37, 40, 128, 0, 0), (130, 16, 16, 384, 0, 1), (256, 131, 132, 256, 64, 1), (96, 5, 8, 128, 0, 2),
    # the small-layer schedule (conv_rowtile.hip: 128 / 256 / 259 / 384 input channels): ragged rows, unaligned rows (ldx = cin = 259),
    # every activation mode incl. raw accumulators
    (4096, 256, 256, 512, 0, 1), (4096, 259, 259, 256, 0, 1), (100, 259, 260, 128, 0, 0), (16384, 128, 128, 128, 0, 2),
    (16384, 256, 256, 128, 0, 1), (1000, 384, 384, 256, 0, 1), (33, 128, 132, 256, 0, 2), (31, 256, 256, 128, 0, 0)])
def test_conv1x1_packed_equals_conv1x1_bitwise(dev, rows, cin, ldx, cout, pool, act):
    """csrc/conv_packed.hip (wave-independent, packed weights) against the workgroup-tiled ancsh_conv1x1: identical bits,
    including ragged row counts, odd k, pooling and a chain continued from acc_init."""
    from articulated_pose_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(rows + cin)
    x = torch.randn(rows, ldx, generator=g).to(dev)
    W = (torch.randn(cin, cout, generator=g) / cin ** 0.5).to(dev)
    bias, scale, shift = [torch.randn(cout, generator=g).to(dev) for _ in range(3)]
    pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(cin, cout), device=dev)
    _lib.call("ancsh_sa_pack_weights", cin, cout, _lib.ptr(W), _lib.ptr(pk))
    orows = rows // pool if pool else rows
    for init_rows in (0, 32 if pool == 0 else 0):
        init = torch.randn((rows + init_rows - 1) // init_rows, cout, generator=g).to(dev) if init_rows else None
        want = torch.full((orows, cout), float("nan"), device=dev); got = want.clone()
        _lib.call("ancsh_conv1x1_ex", rows, cin, cout, _lib.ptr(x), ldx, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift),
                  act, _lib.ptr(want), cout, pool, _lib.ptr(init), init_rows)
        _lib.call("ancsh_conv1x1_packed", rows, cin, cout, _lib.ptr(x), ldx, _lib.ptr(pk), _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift),
                  act, _lib.ptr(got), cout, pool, _lib.ptr(init), init_rows)
        assert not torch.isnan(got).any()
        assert torch.equal(got, want), (init_rows, (got - want).abs().max().item())
    with pytest.raises(ValueError):
        _lib.call("ancsh_conv1x1_packed", rows, cin, 96, _lib.ptr(x), ldx, _lib.ptr(pk), _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift),
                  act, _lib.ptr(got), 96, 0, None, 0)


def test_packed_conv_network_equals_plain_bitwise(dev):
    from articulated_pose_amd import tf_util
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    w = synthetic_weights(3, seed=31)
    P = synth_cloud(np.random.RandomState(4), 32, 1024)
    net = Network(3, w, "ancsh", dev)
    prev = tf_util.PACKED_CONV
    try:
        tf_util.PACKED_CONV = True
        a = {k: v.clone() for k, v in net.predict(P).items()}
        tf_util.PACKED_CONV = False
        b = {k: v.clone() for k, v in net.predict(P).items()}
    finally:
        tf_util.PACKED_CONV = prev
    for k in b:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("K,nocs_type", [(3, "ancsh"), (3, "npcs"), (4, "ancsh"), (2, "npcs")])
def test_fused_tail_equals_layerwise_bitwise(dev, K, nocs_type):
    """csrc/chain.hip (fa_layer3 + fc1 + every head as one launch, activations in LDS) vs one launch per layer."""
    from articulated_pose_amd import architecture
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    mixed = nocs_type == "ancsh"
    w = synthetic_weights(K, mixed_pred=mixed, early_split_nocs=mixed, seed=11)
    P = synth_cloud(np.random.RandomState(K), 2, 1024)
    net = Network(K, w, nocs_type, dev)
    try:
        architecture.FUSED_TAIL = True
        fused = {k: v.clone() for k, v in net.predict(P).items()}
        architecture.FUSED_TAIL = False
        plain = {k: v.clone() for k, v in net.predict(P).items()}
    finally:
        architecture.FUSED_TAIL = True
    assert set(fused) == set(plain)
    for k in plain:
        assert torch.equal(fused[k], plain[k]), k


@pytest.mark.parametrize("b,n,npoint", [(1, 200, 5), (3, 150, 3), (2, 96, 1), (8, 150, 4), (16, 130, 6)])
def test_fused_sa_ragged_group_counts_bitwise(dev, b, n, npoint):
    """Fused SA kernels when the number of neighbourhoods is not a multiple of the 4 (SA1) / 2 (SA2) a workgroup handles, and (b a
    multiple of 8) under the XCD-aware workgroup -> neighbourhood map, incl. a level where only SA2 can use it (npoint = 6)."""
    from articulated_pose_amd import pointnet_util, tf_util
    from articulated_pose_amd.weights import synthetic_weights
    w = synthetic_weights(3, seed=17)
    tf_util.set_variables(w)
    rng = np.random.RandomState(b * 100 + npoint)
    xyz = torch.from_numpy(rng.rand(b, n, 3).astype(np.float32)).to(dev)
    feats = torch.from_numpy(rng.randn(b, n, 128).astype(np.float32)).to(dev)
    outs = []
    for fused in (True, False):
        pointnet_util.FUSED_SA = fused
        try:
            with tf_util.variable_scope("SPFN"), tf_util.variable_scope("est_net"):
                _, p1, _ = pointnet_util.pointnet_sa_module(xyz, None, npoint, 0.4, 64, [64, 64, 128], None, False, False, None, "layer1")
                _, p2, _ = pointnet_util.pointnet_sa_module(xyz, feats, npoint, 0.6, 64, [128, 128, 256], None, False, False, None, "layer2")
        finally:
            pointnet_util.FUSED_SA = True
        outs.append((p1.clone(), p2.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0].shape == (b, npoint, 128) and outs[0][1].shape == (b, npoint, 256)


@pytest.mark.parametrize("K,N,B", [(3, 1024, 8), (2, 2048, 3), (4, 2048, 2), (3, 1000, 1), (3, 777, 5)])
def test_paired_networks_equal_separate_forwards(dev, K, N, B):
    """paired.PairedNetworks: every backbone layer of the ANCSH and the NPCS network in ONE grouped launch (ancsh_*_grouped on
    stacked activations, shared geometry).  All outputs bit-identical to each network's own forward; B = 8 takes the XCD-aware
    workgroup map (16 stacked clouds), the others the plain one; N = 1000 / 777 are ragged row counts."""
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.paired import PairedNetworks
    from articulated_pose_amd.pointnet_util import Geometry
    from articulated_pose_amd.weights import synthetic_weights
    P = torch.from_numpy(synth_cloud(np.random.RandomState(K * 10 + B), B, N)).to(dev)
    a = Network(K, synthetic_weights(K, seed=0), "ancsh", dev)
    n = Network(K, synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1), "npcs", dev)
    alone = [a.predict(P), n.predict(P)]
    pair = PairedNetworks([a, n])
    assert pair.eligible()
    for geometry in (None, Geometry()):
        got = pair.predict(P, geometry)
        for g in range(2):
            assert set(got[g]) == set(alone[g])
            for k in alone[g]:
                assert torch.equal(got[g][k], alone[g][k]), (g, k, float((got[g][k] - alone[g][k]).abs().max()))
    # a store whose backbone is not the ANCSH one must not reach the grouped launches (they read packed weights of fixed shapes)
    odd = dict(a.weights)
    odd["SPFN/est_net/layer3/conv1/weights"] = np.zeros((1, 1, 256, 500), np.float32)
    assert not PairedNetworks([Network(K, odd, "ancsh", dev), n]).eligible()
    # one network "paired" with itself twice, and three networks at once
    tri = PairedNetworks([n, a, n]).predict(P)
    for k in alone[1]:
        assert torch.equal(tri[0][k], alone[1][k]) and torch.equal(tri[2][k], alone[1][k]), k
    for k in alone[0]:
        assert torch.equal(tri[1][k], alone[0][k]), k


def test_grouped_conv_equals_plain_bitwise(dev):
    """ancsh_conv1x1_packed_grouped / ancsh_conv1x1_grouped against one plain call per group: the small-layer schedule
    (conv_rowtile, with and without acc_init, raw accumulators), the wave-independent kernel with pooling, the few-rows product."""
    import ctypes
    from articulated_pose_amd import _lib
    rng = np.random.RandomState(0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    tab = lambda ts: ctypes.cast((ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), ctypes.c_void_p)
    for G, rows, cin, cout, pool, act, init_rows in ((2, 4096, 256, 256, 0, 1, 0), (2, 1000, 259, 256, 0, 1, 0), (3, 512, 128, 128, 0, 2, 0),
                                                     (2, 2048, 256, 256, 0, 1, 128), (2, 4096, 512, 1024, 128, 1, 0), (4, 640, 384, 256, 0, 0, 0)):
        ldx = (cin + 3) // 4 * 4 if cin != 259 else 259
        x = T(rng.randn(G * rows, ldx))
        W = [T(rng.randn(cin, cout) / np.sqrt(cin)) for _ in range(G)]
        pk = []
        for w in W:
            p = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(cin, cout), dtype=torch.float32, device=dev)
            _lib.call("ancsh_sa_pack_weights", cin, cout, _lib.ptr(w), _lib.ptr(p))
            pk.append(p)
        b, sc, sh = ([T(rng.randn(cout)) for _ in range(G)] for _ in range(3))
        init = T(rng.randn(G * rows // init_rows, cout)) if init_rows else None
        orows = rows // pool if pool else rows
        want = torch.empty((G * orows, cout), dtype=torch.float32, device=dev)
        for g in range(G):
            _lib.call("ancsh_conv1x1_packed", rows, cin, cout, _lib.ptr(x[g * rows:]), ldx, _lib.ptr(pk[g]), _lib.ptr(b[g]), _lib.ptr(sc[g]),
                      _lib.ptr(sh[g]), act, _lib.ptr(want[g * orows:]), cout, pool, _lib.ptr(None if init is None else init[g * rows // init_rows:]), init_rows)
        got = torch.full_like(want, float("nan"))
        _lib.call("ancsh_conv1x1_packed_grouped", G, rows, cin, cout, _lib.ptr(x), ldx, tab(pk), None if act == 2 else tab(b),
                  None if act == 2 else tab(sc), None if act == 2 else tab(sh), act, _lib.ptr(got), cout, pool, _lib.ptr(init), init_rows)
        assert torch.equal(got, want), (G, rows, cin, cout, pool, act, init_rows)
    G, rows, cin, cout = 2, 32, 1024, 256
    x, W = T(rng.randn(G * rows, cin)), [T(rng.randn(cin + 256, cout) / 32) for _ in range(G)]
    want, got = torch.empty((G * rows, cout), dtype=torch.float32, device=dev), torch.empty((G * rows, cout), dtype=torch.float32, device=dev)
    for g in range(G):
        _lib.call("ancsh_conv1x1", rows, cin, cout, _lib.ptr(x[g * rows:]), cin, _lib.ptr(W[g]), None, None, None, 2, _lib.ptr(want[g * rows:]), cout, 0)
    _lib.call("ancsh_conv1x1_grouped", G, rows, cin, cout, _lib.ptr(x), cin, tab(W), None, None, None, 2, _lib.ptr(got), cout, 0)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        _lib.call("ancsh_conv1x1_packed_grouped", 5, 64, 128, 128, _lib.ptr(x), 128, tab(pk), tab(b), tab(sc), tab(sh), 1, _lib.ptr(got), 128, 0, None, 0)
