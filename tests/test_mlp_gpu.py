"""GPU parity of the shared-MLP layer (f32 MFMA conv1x1 + bias + folded BN + ReLU [+ max-pool]) and
of the head activation kernel against the CPU oracle.  The GEMM accumulates the same k-ordered fmaf
chain as the oracle, so the comparison is bit-exact; a float64 torch reference bounds the error of
both against exact arithmetic."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_layer(rng, cin, cout, bn=True):
    lim = np.sqrt(6.0 / (cin + cout))
    w = rng.uniform(-lim, lim, (cin, cout)).astype(np.float32)
    b = (0.1 * rng.randn(cout)).astype(np.float32)
    if bn:
        gamma = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        beta = (0.1 * rng.randn(cout)).astype(np.float32)
        mean = (0.1 * rng.randn(cout)).astype(np.float32)
        var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        scale = (gamma * (np.float32(1.0) / np.sqrt(var + np.float32(1e-3)))).astype(np.float32)
        shift = (beta - mean * scale).astype(np.float32)
    else:
        scale, shift = np.ones(cout, np.float32), np.zeros(cout, np.float32)
    return dict(w=w, b=b, scale=scale, shift=shift)


def run_gpu(x, layer, act, dev, pool=0, ldx=None):
    from articulated_pose_amd import _lib
    rows, cin = x.shape
    cout = layer["w"].shape[1]
    ldx = ldx or cin
    xb = torch.zeros((rows, ldx), device=dev)
    xb[:, :cin] = torch.from_numpy(x).to(dev)
    t = {k: torch.from_numpy(v).to(dev) for k, v in layer.items()}
    orow = rows // pool if pool else rows
    y = torch.full((orow, cout), float("nan"), device=dev)
    _lib.call("ancsh_conv1x1", rows, cin, cout, _lib.ptr(xb), ldx, _lib.ptr(t["w"]), _lib.ptr(t["b"]),
              _lib.ptr(t["scale"]), _lib.ptr(t["shift"]), act, _lib.ptr(y), cout, pool)
    return y.cpu().numpy()


@pytest.mark.parametrize("rows,cin,cout", [(128, 3, 64), (256, 64, 64), (384, 64, 128), (128, 131, 128),
                                           (200, 128, 256), (128, 259, 256), (130, 1280, 256), (64, 512, 1024),
                                           (1000, 128, 16), (257, 128, 9), (129, 128, 10), (77, 5, 3)])
@pytest.mark.parametrize("act", [0, 1])
def test_conv1x1_bit_exact(oracle, dev, rows, cin, cout, act):
    rng = np.random.RandomState(rows + cin + cout)
    x = rng.randn(rows, cin).astype(np.float32)
    layer = make_layer(rng, cin, cout, bn=bool(act))
    want = oracle.conv1x1(x, layer, act)
    got = run_gpu(x, layer, act, dev)
    np.testing.assert_array_equal(got, want)
    # float64 reference: both within f32 round-off of exact arithmetic
    ref = (x.astype(np.float64) @ layer["w"].astype(np.float64) + layer["b"]) * layer["scale"] + layer["shift"]
    if act:
        ref = np.maximum(ref, 0)
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_conv1x1_padded_rows(oracle, dev):
    """Row stride > cin with 16-byte aligned rows takes the float4 staging path (131 -> ld 132)."""
    rng = np.random.RandomState(3)
    x = rng.randn(256, 131).astype(np.float32)
    layer = make_layer(rng, 131, 128)
    np.testing.assert_array_equal(run_gpu(x, layer, 1, dev, ldx=132), oracle.conv1x1(x, layer, 1))


@pytest.mark.parametrize("pool,cout", [(64, 128), (64, 256), (64, 64), (128, 1024), (128, 256)])
def test_conv1x1_fused_maxpool(oracle, dev, pool, cout):
    rng = np.random.RandomState(pool + cout)
    rows, cin = pool * 6, 96
    x = rng.randn(rows, cin).astype(np.float32)
    layer = make_layer(rng, cin, cout)
    want = oracle.group_max(oracle.conv1x1(x, layer, 1).reshape(rows // pool, pool, cout))
    np.testing.assert_array_equal(run_gpu(x, layer, 1, dev, pool=pool), want)


def test_group_max(oracle, dev):
    from articulated_pose_amd import _lib
    x = np.random.RandomState(0).randn(10, 64, 48).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    y = torch.empty((10, 48), device=dev)
    _lib.call("ancsh_group_max", 10, 64, 48, _lib.ptr(xt), _lib.ptr(y))
    np.testing.assert_array_equal(y.cpu().numpy(), oracle.group_max(x))


@pytest.mark.gpu
def test_pack_weights_layout_and_errors(dev):
    """ancsh_sa_pack_weights against the layout the header documents, for n not a multiple of 32 and odd k."""
    from articulated_pose_amd import _lib
    for k, n in ((131, 128), (3, 64), (128, 9), (7, 33)):
        W = torch.randn(k, n, device=dev)
        nf = _lib.lib().ancsh_sa_packed_weight_floats(k, n)
        pk = torch.full((nf,), float("nan"), device=dev)
        _lib.call("ancsh_sa_pack_weights", k, n, _lib.ptr(W), _lib.ptr(pk))
        tn, ns = (n + 31) // 32, ((k + 1) // 2 + 3) // 4
        got = pk.cpu().numpy().reshape(ns, tn, 64, 4)
        Wc = W.cpu().numpy()
        want = np.zeros_like(got)
        for slot in range(ns):
            for q in range(4):
                for half in range(2):
                    kk = 2 * (4 * slot + q) + half
                    if kk < k:
                        for j in range(tn):
                            c0, c1 = j * 32, min(n, j * 32 + 32)
                            want[slot, j, half * 32:half * 32 + (c1 - c0), q] = Wc[kk, c0:c1]
        np.testing.assert_array_equal(got, want)
    with pytest.raises(ValueError):
        _lib.call("ancsh_sa_pack_weights", 0, 32, _lib.ptr(W), _lib.ptr(pk))
    with pytest.raises(ValueError):
        _lib.call("ancsh_sa_pack_weights", 4, 32, None, _lib.ptr(pk))


@pytest.mark.parametrize("rows,cin,cout,ldx", [(32, 1024, 256, 1024), (1, 1024, 256, 1024), (7, 100, 60, 104), (64, 64, 64, 64),
                                               (33, 3, 5, 4), (16, 130, 257, 132)])
def test_conv1x1_few_rows_raw_accumulators(oracle, dev, rows, cin, cout, ldx):
    """ANCSH_ACT_RAW on <= 64 rows (the per-cloud partial product of the single-source FP module): the VALU fmaf-chain kernel
    returns the raw k-ordered accumulators -- bit-equal to the oracle's chain with zero bias / unit scale."""
    rng = np.random.RandomState(rows * 3 + cin + cout)
    x = rng.randn(rows, cin).astype(np.float32)
    w = (rng.randn(cin, cout) / np.sqrt(cin)).astype(np.float32)
    raw = dict(w=w, b=np.zeros(cout, np.float32), scale=np.ones(cout, np.float32), shift=np.zeros(cout, np.float32))
    want = oracle.conv1x1(x, raw, 0)          # (acc + 0) * 1 + 0 == acc exactly
    got = run_gpu(x, raw, 2, dev, ldx=ldx)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("rows,cin,ldx,offset", [(1, 131, 132, 0), (31, 131, 132, 0), (33, 131, 131, 0), (100, 128, 128, 0),
                                                 (257, 131, 132, 1), (1000, 131, 136, 0)])
def test_mlp_chain_program_equals_layer_by_layer(dev, rows, cin, ldx, offset):
    """ancsh_mlp_chain on ragged row counts (not a multiple of the 32-row wave tile), 16-byte aligned and unaligned inputs
    (odd row stride / base offset by one float), both activations, LDS and global destinations, an in-place layer and head
    blocks of 7 and 32 columns: every destination is bit-equal to the same layers run one ancsh_conv1x1 launch at a time."""
    import ctypes
    from articulated_pose_amd import _lib
    rng = np.random.RandomState(rows)
    k0 = 131 if cin == 131 else 128
    x = rng.randn(rows, cin).astype(np.float32)
    store = torch.zeros(rows * ldx + 8, device=dev)
    xb = store[offset:offset + rows * ldx].view(rows, ldx)
    xb[:, :cin] = torch.from_numpy(x).to(dev)
    if k0 > cin:
        pytest.skip("first layer must consume cin")
    specs = [(k0, 128, 1, 0, 1, None), (128, 128, 0, 1, 0, None), (128, 7, 0, 0, -1, 0), (128, 128, 1, 0, 0, None),
             (128, 32, 1, 0, -1, 8), (128, 128, 1, 1, 1, None), (128, 13, 0, 1, -1, 40)]
    ld_out = 56
    logits = torch.full((rows, ld_out), float("nan"), device=dev)
    layers, ops, ptrs, keep = [], [], [], []
    for (k, n, act, src, dst, col) in specs:
        L = {kk: torch.from_numpy(v).to(dev) for kk, v in make_layer(rng, k, n).items()}
        pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(k, n), device=dev)
        _lib.call("ancsh_sa_pack_weights", k, n, _lib.ptr(L["w"]), _lib.ptr(pk))
        out = None if col is None else logits[:, col:]
        ops += [k, n, act, src, dst, ld_out if out is not None else 0]
        ptrs += [_lib.ptr(pk), _lib.ptr(L["b"]), _lib.ptr(L["scale"]), _lib.ptr(L["shift"]), _lib.ptr(out)]
        layers.append(L); keep.append(pk)
    c_ops = (ctypes.c_int * len(ops))(*ops)
    c_ptrs = (ctypes.c_void_p * len(ptrs))(*ptrs)
    _lib.call("ancsh_mlp_chain", rows, cin, _lib.ptr(xb), ldx, len(specs), ctypes.cast(c_ops, ctypes.c_void_p), ctypes.cast(c_ptrs, ctypes.c_void_p))
    # the same program, one launch per layer
    tiles = [torch.zeros((rows, 131), device=dev), torch.zeros((rows, 131), device=dev)]
    tiles[0][:, :cin] = xb[:, :cin]
    want = torch.full((rows, ld_out), float("nan"), device=dev)
    for (k, n, act, src, dst, col), L in zip(specs, layers):
        y = torch.empty((rows, n), device=dev)
        _lib.call("ancsh_conv1x1", rows, k, n, _lib.ptr(tiles[src]), 131, _lib.ptr(L["w"]), _lib.ptr(L["b"]), _lib.ptr(L["scale"]),
                  _lib.ptr(L["shift"]), act, _lib.ptr(y), n, 0)
        if col is None:
            tiles[dst] = torch.zeros((rows, 131), device=dev)
            tiles[dst][:, :n] = y
        else:
            want[:, col:col + n] = y
    got, ref = logits.cpu().numpy(), want.cpu().numpy()
    written = ~np.isnan(ref)
    assert written.sum() == rows * (7 + 32 + 13)
    assert np.array_equal(np.isnan(got), np.isnan(ref))          # nothing outside the head blocks' columns is touched
    assert np.array_equal(got[written].view(np.uint32), ref[written].view(np.uint32))


@pytest.mark.parametrize("rows,ldx", [(1000, 132), (33, 131), (4096, 136)])
def test_mlp_chain_grouped_one_tile_equals_layer_by_layer(dev, rows, ldx):
    """ancsh_mlp_chain_grouped: TWO networks' programs in one launch, one LDS tile per wave (every layer in place), the trunk saved
    to / restored from the scratch rows for its second 128-wide consumer -- every head block bit-equal to the same layers run one
    ancsh_conv1x1 launch at a time.  Program A has the ANCSH tail's shape (save at op 3, restore at op 8), program B none."""
    import ctypes
    from articulated_pose_amd import _lib
    rng = np.random.RandomState(rows)
    SAVE, RESTORE = 1, 2
    # (k, n, act, flags, out column or None, source: 'tile' / 'saved')
    prog_a = [(131, 128, 1, 0, None), (128, 128, 1, 0, None), (128, 128, 1, 0, None), (128, 128, 1, SAVE, None), (128, 7, 0, 0, 0),
              (128, 32, 0, 0, 8), (128, 128, 0, 0, None), (128, 9, 0, 0, 40), (128, 128, 1, RESTORE, None), (128, 128, 1, 0, None), (128, 10, 0, 0, 52)]
    prog_b = [(131, 128, 1, 0, None), (128, 128, 0, 0, None), (128, 13, 0, 0, 0), (128, 128, 1, 0, None), (128, 10, 1, 0, 16)]
    ld_out = 64
    x = rng.randn(2 * rows, 131).astype(np.float32)
    store = torch.zeros((2 * rows, ldx), device=dev)
    store[:, :131] = torch.from_numpy(x).to(dev)
    logits = [torch.full((rows, ld_out), float("nan"), device=dev) for _ in range(2)]
    want = [torch.full((rows, ld_out), float("nan"), device=dev) for _ in range(2)]
    scratch = torch.empty((2 * rows, 128), device=dev)
    all_ops, all_ptrs, keep = [], [], []
    for g, prog in enumerate((prog_a, prog_b)):
        ops, ptrs = [], []
        tile = torch.zeros((rows, 131), device=dev)
        tile[:] = store[g * rows:(g + 1) * rows, :131]
        saved = None
        for (k, n, act, flags, col) in prog:
            L = {kk: torch.from_numpy(v).to(dev) for kk, v in make_layer(rng, k, n).items()}
            pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(k, n), device=dev)
            _lib.call("ancsh_sa_pack_weights", k, n, _lib.ptr(L["w"]), _lib.ptr(pk))
            out = None if col is None else logits[g][:, col:]
            ops += [k, n, act, flags, ld_out if out is not None else 0]
            ptrs += [_lib.ptr(pk), _lib.ptr(L["b"]), _lib.ptr(L["scale"]), _lib.ptr(L["shift"]), _lib.ptr(out)]
            keep += [L, pk]
            # the same layer on its own
            if flags & RESTORE:
                tile = torch.zeros((rows, 131), device=dev)
                tile[:, :128] = saved
            y = torch.empty((rows, n), device=dev)
            _lib.call("ancsh_conv1x1", rows, k, n, _lib.ptr(tile), 131, _lib.ptr(L["w"]), _lib.ptr(L["b"]), _lib.ptr(L["scale"]), _lib.ptr(L["shift"]),
                      act, _lib.ptr(y), n, 0)
            if col is None:
                tile = torch.zeros((rows, 131), device=dev)
                tile[:, :n] = y
                if flags & SAVE:
                    saved = y.clone()
            else:
                want[g][:, col:col + n] = y
        all_ops.append((ctypes.c_int * len(ops))(*ops))
        all_ptrs.append((ctypes.c_void_p * len(ptrs))(*ptrs))
    nops = (ctypes.c_int * 2)(len(prog_a), len(prog_b))
    ops_tab = (ctypes.c_void_p * 2)(*[ctypes.cast(o, ctypes.c_void_p) for o in all_ops])
    ptr_tab = (ctypes.c_void_p * 2)(*[ctypes.cast(o, ctypes.c_void_p) for o in all_ptrs])
    _lib.call("ancsh_mlp_chain_grouped", 2, rows, 131, _lib.ptr(store), ldx, ctypes.cast(nops, ctypes.c_void_p), ctypes.cast(ops_tab, ctypes.c_void_p),
              ctypes.cast(ptr_tab, ctypes.c_void_p), _lib.ptr(scratch))
    for g in range(2):
        got, ref = logits[g].cpu().numpy(), want[g].cpu().numpy()
        written = ~np.isnan(ref)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.array_equal(got[written].view(np.uint32), ref[written].view(np.uint32)), g
    with pytest.raises(ValueError):      # a restore before any save is a program error
        bad = (ctypes.c_int * 5)(128, 128, 1, RESTORE, 0)
        _lib.call("ancsh_mlp_chain_grouped", 1, rows, 131, _lib.ptr(store), ldx, ctypes.cast((ctypes.c_int * 1)(1), ctypes.c_void_p),
                  ctypes.cast((ctypes.c_void_p * 1)(ctypes.cast(bad, ctypes.c_void_p)), ctypes.c_void_p),
                  ctypes.cast((ctypes.c_void_p * 1)(ctypes.cast(all_ptrs[0], ctypes.c_void_p)), ctypes.c_void_p), _lib.ptr(scratch))
