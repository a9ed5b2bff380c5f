"""The middle of the backbone (layer3, fa_layer1, fa_layer2) as chain launches (csrc/mid_chain.hip) against the layer-by-layer
launches of rounds 3-4 -- pointnet_plusplus/architectures.py:72-82 -- bit for bit, level by level and end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(dev, K=3):
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    a = Network(K, synthetic_weights(K, seed=0), "ancsh", dev)
    n = Network(K, synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1), "npcs", dev)
    return a, n


def _inputs(dev, G, B, seed):
    """random stand-ins for the mid-section's inputs with the backbone's shapes; the 3-NN geometry from the real operator"""
    from articulated_pose_amd.tf_ops import tf_interpolate
    rng = np.random.RandomState(seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    l1_xyz, l2_xyz = T(rng.uniform(-1, 1, (B, 512, 3))), T(rng.uniform(-1, 1, (B, 128, 3)))
    l2_points = T(np.abs(rng.randn(G * B, 128, 256)))
    l1_points = T(np.abs(rng.randn(G * B, 512, 128)))
    _dist, idx, w = tf_interpolate.three_nn_weights(l1_xyz, l2_xyz)
    return l2_xyz, l2_points, l1_points, idx.contiguous(), w.contiguous()


@pytest.mark.parametrize("G,B", [(1, 1), (2, 3), (2, 8), (3, 5), (4, 2), (2, 16), (2, 32), (3, 24)])
def test_mid_chains_equal_layer_by_layer(dev, G, B):
    """`_mid_chains` (4 launches) == `_mid_layers` (9 conv launches + concat + interpolate): torch.equal on the level's output.
    (2, 8), (2, 16), (2, 32), (3, 24) take the XCD-aware tile map of the fa_layer2 chain, odd B the row blocks that end inside a network;
    (2, 32) is the benchmark's shape (256 row tiles of layer3 = one workgroup per CU)."""
    from articulated_pose_amd.paired import PairedNetworks
    a, n = _nets(dev)
    pair = PairedNetworks(([a, n] * 2)[:G])
    l2_xyz, l2_points, l1_points, fi2, fw2 = _inputs(dev, G, B, 10 * G + B)
    L3 = [pair._layers("layer3/conv%d" % i) for i in range(3)]
    F1 = [pair._layers("fa_layer1/conv_%d" % i) for i in range(2)]
    F2 = [pair._layers("fa_layer2/conv_%d" % i) for i in range(2)]
    want = pair._mid_layers(B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2)
    got = pair._mid_chains(B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2)
    torch.cuda.synchronize()
    assert want.shape == got.shape == (G * B * 512, 128)
    assert torch.equal(want, got), float((want - got).abs().max())
    assert float(got.abs().max()) > 0


def test_each_chain_against_its_layers(dev):
    """level by level, so that a mismatch names its kernel: layer3's tile maxima, the single-source product, fa_layer1, fa_layer2"""
    import ctypes
    from articulated_pose_amd import _lib, tf_util
    from articulated_pose_amd.paired import PairedNetworks, _table
    a, n = _nets(dev)
    G, B = 2, 5
    pair = PairedNetworks([a, n])
    l2_xyz, l2_points, l1_points, fi2, fw2 = _inputs(dev, G, B, 77)
    f = dict(dtype=torch.float32, device=dev)
    L3 = [pair._layers("layer3/conv%d" % i) for i in range(3)]
    F1 = [pair._layers("fa_layer1/conv_%d" % i) for i in range(2)]
    F2 = [pair._layers("fa_layer2/conv_%d" % i) for i in range(2)]

    def params(levels, row0s):
        return _table([_lib.ptr(v) for g in range(G) for ls, r0 in zip(levels, row0s)
                       for v in (tf_util.packed_weight(ls[g], r0), ls[g]["b"], ls[g]["scale"], ls[g]["shift"])])

    # layer3
    x3 = torch.cat([l2_xyz.unsqueeze(0).expand(G, B, 128, 3).reshape(G * B, 128, 3), l2_points], dim=2)
    h = pair._conv(L3[0], x3, B * 128, 259, 259, 256)
    h = pair._conv(L3[1], h, B * 128, 256, 256, 512)
    l3 = pair._conv(L3[2], h, B * 128, 512, 512, 1024, pool=128)
    tile_max = torch.full((G * B, 4, 1024), float("nan"), **f)
    p3 = params(L3, (0, 0, 0))
    _lib.call("ancsh_sa3_chain_grouped", G, B, 128, 256, 256, 512, 1024, _lib.ptr(l2_xyz), _lib.ptr(l2_points), p3.p, _lib.ptr(tile_max))
    assert torch.equal(tile_max.max(dim=1).values, l3)
    # single-source product (also with one part per cloud = the plain few-rows product)
    w1 = _table([_lib.ptr(l["w"]) for l in F1[0]])
    want = torch.empty((G * B, 256), **f)
    _lib.call("ancsh_conv1x1_grouped", G, B, 1024, 256, _lib.ptr(l3), 1024, w1.p, None, None, None, 2, _lib.ptr(want), 256, 0)
    pairs = torch.stack([tile_max[:, :2].max(dim=1).values, tile_max[:, 2:].max(dim=1).values], dim=1).contiguous()     # generic nparts path
    triple = torch.stack([tile_max[:, 0], tile_max[:, 1], tile_max[:, 2:].max(dim=1).values], dim=1).contiguous()
    for x, nparts in ((tile_max, 4), (l3, 1), (pairs, 2), (triple, 3)):
        init = torch.full((G * B, 256), float("nan"), **f)
        _lib.call("ancsh_fp_single_source_init", G, B, 1024, 256, nparts, _lib.ptr(x), w1.p, _lib.ptr(init))
        assert torch.equal(init, want), nparts
    # fa_layer1
    h = pair._conv(F1[0], l2_points, B * 128, 256, 256, 256, acc_init=want, init_rows=128, row0=1024)
    l2_up = pair._conv(F1[1], h, B * 128, 256, 256, 256)
    got = torch.full((G * B * 128, 256), float("nan"), **f)
    p1 = params(F1, (1024, 0))
    _lib.call("ancsh_fp1_chain_grouped", G, B, 128, 256, 256, 256, _lib.ptr(l2_points), _lib.ptr(want), p1.p, _lib.ptr(got))
    assert torch.equal(got, l2_up)
    # fa_layer2
    buf = torch.empty((G * B, 512, 384), **f)
    _lib.call("ancsh_fp_interpolate_concat_ex", G * B, 128, 256, 512, _lib.ptr(l2_up), _lib.ptr(fi2), _lib.ptr(fw2), _lib.ptr(l1_points), 128,
              _lib.ptr(buf), 384, B, G * B)
    h = pair._conv(F2[0], buf, B * 512, 384, 384, 256)
    l1_up = pair._conv(F2[1], h, B * 512, 256, 256, 128)
    got = torch.full((G * B * 512, 128), float("nan"), **f)
    p2 = params(F2, (0, 0))
    _lib.call("ancsh_fp2_chain_grouped", G, B, 128, 512, 256, 128, 256, 128, _lib.ptr(l2_up), _lib.ptr(fi2), _lib.ptr(fw2), _lib.ptr(l1_points),
              p2.p, _lib.ptr(got))
    assert torch.equal(got, l1_up)


def test_paired_forward_same_bits_with_and_without_mid_chains(dev, monkeypatch):
    """the whole paired forward with ANCSH_MID_CHAIN on (default) and off: every head tensor of both networks torch.equal"""
    from articulated_pose_amd import paired
    from test_network_gpu import synth_cloud
    a, n = _nets(dev)
    P = torch.from_numpy(synth_cloud(np.random.RandomState(5), 8, 1024)).to(dev)
    pair = paired.PairedNetworks([a, n])
    assert paired.MID_CHAIN
    on = pair.predict(P)
    monkeypatch.setattr(paired, "MID_CHAIN", False)
    off = pair.predict(P)
    for g in range(2):
        for k in on[g]:
            assert torch.equal(on[g][k], off[g][k]), (g, k)


@pytest.mark.parametrize("K,N,B", [(3, 1024, 8), (2, 2048, 3), (4, 2048, 2), (3, 1024, 1), (3, 1000, 2)])
def test_tail_chain_with_interpolation_in_the_load_same_bits(dev, monkeypatch, K, N, B):
    """ancsh_mlp_chain_grouped_fp (fa_layer3's three_interpolate + concat built in the chain's tile load, XCD-aware tile map at B = 8)
    against the materialised concat buffer + ancsh_mlp_chain_grouped: every head tensor of both networks torch.equal; N = 1000 is not
    a multiple of 128 and must take the materialised path by itself."""
    from articulated_pose_amd import paired
    from test_network_gpu import synth_cloud
    a, n = _nets(dev, K)
    P = torch.from_numpy(synth_cloud(np.random.RandomState(K + N + B), B, N)).to(dev)
    pair = paired.PairedNetworks([a, n])
    assert paired.TAIL_FP
    on = pair.predict(P)
    monkeypatch.setattr(paired, "TAIL_FP", False)
    off = pair.predict(P)
    for g in range(2):
        for k in on[g]:
            assert torch.equal(on[g][k], off[g][k]), (g, k)
    one = paired.PairedNetworks([a]).predict(P)[0]             # one network in the launch
    monkeypatch.setattr(paired, "TAIL_FP", True)
    one_fp = paired.PairedNetworks([a]).predict(P)[0]
    for k in one:
        assert torch.equal(one[k], one_fp[k]) and torch.equal(one[k], on[0][k]), k


@pytest.mark.parametrize("G,B,npts,n1", [(2, 3, 64, 96), (1, 5, 32, 32), (3, 2, 64, 160)])
def test_chains_at_other_level_sizes(dev, G, B, npts, n1):
    """The ABI promises any npts % 32 == 0 (layer3 / fa_layer1) and any n % 32 == 0 (fa_layer2), not only the backbone's 128 / 512:
    64- and 32-point levels (pool sizes the layer-by-layer kernels serve) and 96 / 32 / 160 interpolation targets, against the
    layer-by-layer calls."""
    from articulated_pose_amd import _lib, tf_util
    from articulated_pose_amd.paired import PairedNetworks, _table
    from articulated_pose_amd.tf_ops import tf_interpolate
    a, n = _nets(dev)
    pair = PairedNetworks(([a, n] * 2)[:G])
    rng = np.random.RandomState(npts + n1)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    f = dict(dtype=torch.float32, device=dev)
    l1_xyz, l2_xyz = T(rng.uniform(-1, 1, (B, n1, 3))), T(rng.uniform(-1, 1, (B, npts, 3)))
    l2_points, l1_points = T(np.abs(rng.randn(G * B, npts, 256))), T(np.abs(rng.randn(G * B, n1, 128)))
    _d, fi2, fw2 = tf_interpolate.three_nn_weights(l1_xyz, l2_xyz)
    L3 = [pair._layers("layer3/conv%d" % i) for i in range(3)]
    F1 = [pair._layers("fa_layer1/conv_%d" % i) for i in range(2)]
    F2 = [pair._layers("fa_layer2/conv_%d" % i) for i in range(2)]

    def params(levels, row0s):
        return _table([_lib.ptr(v) for g in range(G) for ls, r0 in zip(levels, row0s)
                       for v in (tf_util.packed_weight(ls[g], r0), ls[g]["b"], ls[g]["scale"], ls[g]["shift"])])

    # layer by layer (generic kernels; the pooled last layer of layer3 through the plain per-group conv: pool 64 exists, pool 32 does not)
    x3 = torch.cat([l2_xyz.unsqueeze(0).expand(G, B, npts, 3).reshape(G * B, npts, 3), l2_points], dim=2).contiguous()
    h = pair._conv(L3[0], x3, B * npts, 259, 259, 256)
    h = pair._conv(L3[1], h, B * npts, 256, 256, 512)
    h = pair._conv(L3[2], h, B * npts, 512, 512, 1024)
    l3 = h.view(G * B, npts, 1024).max(dim=1).values.contiguous()
    w1 = _table([_lib.ptr(l["w"]) for l in F1[0]])
    init = torch.empty((G * B, 256), **f)
    _lib.call("ancsh_conv1x1_grouped", G, B, 1024, 256, _lib.ptr(l3), 1024, w1.p, None, None, None, 2, _lib.ptr(init), 256, 0)
    h = pair._conv(F1[0], l2_points, B * npts, 256, 256, 256, acc_init=init, init_rows=npts, row0=1024)
    l2_up = pair._conv(F1[1], h, B * npts, 256, 256, 256)
    buf = torch.empty((G * B, n1, 384), **f)
    _lib.call("ancsh_fp_interpolate_concat_ex", G * B, npts, 256, n1, _lib.ptr(l2_up), _lib.ptr(fi2), _lib.ptr(fw2), _lib.ptr(l1_points), 128,
              _lib.ptr(buf), 384, B, G * B)
    h = pair._conv(F2[0], buf, B * n1, 384, 384, 256)
    want = pair._conv(F2[1], h, B * n1, 256, 256, 128)
    # chains
    tile_max = torch.full((G * B, npts // 32, 1024), float("nan"), **f)
    _lib.call("ancsh_sa3_chain_grouped", G, B, npts, 256, 256, 512, 1024, _lib.ptr(l2_xyz), _lib.ptr(l2_points), params(L3, (0, 0, 0)).p, _lib.ptr(tile_max))
    assert torch.equal(tile_max.max(dim=1).values, l3)
    init2 = torch.full((G * B, 256), float("nan"), **f)
    _lib.call("ancsh_fp_single_source_init", G, B, 1024, 256, npts // 32, _lib.ptr(tile_max), w1.p, _lib.ptr(init2))
    assert torch.equal(init2, init)
    up2 = torch.full((G * B * npts, 256), float("nan"), **f)
    _lib.call("ancsh_fp1_chain_grouped", G, B, npts, 256, 256, 256, _lib.ptr(l2_points), _lib.ptr(init2), params(F1, (1024, 0)).p, _lib.ptr(up2))
    assert torch.equal(up2, l2_up)
    got = torch.full((G * B * n1, 128), float("nan"), **f)
    _lib.call("ancsh_fp2_chain_grouped", G, B, npts, n1, 256, 128, 256, 128, _lib.ptr(up2), _lib.ptr(fi2), _lib.ptr(fw2), _lib.ptr(l1_points),
              params(F2, (0, 0)).p, _lib.ptr(got))
    assert torch.equal(got, want)
