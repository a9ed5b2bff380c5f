"""Inference-only counterpart of lib/architecture.py: get_per_point_model_new (:86-161) and
joint_est_model (:195-208).

The ten tiny head convolutions are issued as three GEMM launches on column-concatenated kernels
(every output column is an independent dot product, so this is exact), writing one logits row per
point; a single kernel then applies softmax / sigmoid / tanh and composes gocs (:124-159).
"""
import torch

from . import _lib, tf_util
from .architectures import build_pointnet2_shared


def joint_est_model(scope, X, is_training, bn_decay, n_max_parts=3, pred_joint_ind=False):
    """Returns the RAW (pre-activation) joint_axis, unitvec, heatmap, joint_cls like the reference."""
    layer_dims = [128, 128]
    with tf_util.variable_scope(scope):
        for j, dim in enumerate(layer_dims):
            X = tf_util.conv1d(X, dim, 1, padding='VALID', bn=True, is_training=is_training,
                               scope='fc3_{}'.format(j), bn_decay=bn_decay)
            X = tf_util.dropout(X, keep_prob=0.5, is_training=is_training, scope='dp1')
        joint_axis = tf_util.conv1d(X, 3, 1, padding='VALID', activation_fn=None, scope='fc4_0')
        univect = tf_util.conv1d(X, 3, 1, padding='VALID', activation_fn=None, scope='fc4_1')
        heatmap = tf_util.conv1d(X, 1, 1, padding='VALID', activation_fn=None, scope='fc4_2')
        joint_cls = tf_util.conv1d(X, n_max_parts, 1, padding='VALID', activation_fn=None, scope='fc4_3')
    return joint_axis, univect, heatmap, joint_cls


import os as _os

FUSED_TAIL = _os.environ.get('ANCSH_FUSED_TAIL', '1') != '0'    # fa_layer3 + fc1 + all heads as one ancsh_mlp_chain launch (False = layer-by-layer, same bits)


def _head_dims(K, mixed_pred, early_split_nocs):
    """(out_dims of the nocs_net heads, True when the chain kernel can take them: head blocks of 128 or <= 32 columns)."""
    out_dims = [K, 3 * K] + ([K, 3 * K] if mixed_pred else []) + [1]
    head_widths = ([out_dims[0], sum(out_dims[2:]), out_dims[1]] if early_split_nocs else [sum(out_dims)]) + [10]
    return out_dims, all(n <= 32 for n in head_widths)


def _fused_tail(scope, P, K, mixed_pred, early_split_nocs):
    """build_pointnet2_shared up to fa_layer2, then fa_layer3's interpolation and EVERYTHING after it (three
    fa_layer3 convs, fc1, nocs_net, joint_net) as one kernel.  Returns the logits matrix (rows, ld) in the layout
    ancsh_head_activations expects, or None when the shapes are not the ANCSH ones."""
    from . import pointnet_util as pu
    # the chain kernel takes head blocks of 128 or <= 32 columns: decide BEFORE any backbone kernel is launched
    if not _head_dims(K, mixed_pred, early_split_nocs)[1]:
        return None
    with tf_util.variable_scope('est_net'):
        l0_xyz = P
        l1_xyz, l1_points, _ = pu.pointnet_sa_module(l0_xyz, P[:, :, 3:3], npoint=512, radius=0.2, nsample=64, mlp=[64, 64, 128],
                                                      mlp2=None, group_all=False, is_training=False, bn_decay=None, scope='layer1')
        l2_xyz, l2_points, _ = pu.pointnet_sa_module(l1_xyz, l1_points, npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 256],
                                                      mlp2=None, group_all=False, is_training=False, bn_decay=None, scope='layer2')
        l3_xyz, l3_points, _ = pu.pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                                      mlp=[256, 512, 1024], mlp2=None, group_all=True, is_training=False,
                                                      bn_decay=None, scope='layer3')
        l2_points = pu.pointnet_fp_module(l2_xyz, l3_xyz, l2_points, l3_points, [256, 256], False, None, scope='fa_layer1')
        l1_points = pu.pointnet_fp_module(l1_xyz, l2_xyz, l1_points, l2_points, [256, 128], False, None, scope='fa_layer2')
        with tf_util.variable_scope('fa_layer3'):
            x = pu.fp_interpolate_concat(l0_xyz, l1_xyz, l0_xyz, l1_points)          # (B, N, 132): [interp(128) | xyz(3) | pad]
    return _tail_chain(x, P.shape[0] * P.shape[1], K, mixed_pred, early_split_nocs)


CH_SAVE, CH_RESTORE = 1, 2       # ancsh_mlp_chain_grouped op flags


def _bf16x3_head_params(layer):
    """a head block (128 -> n <= 32) for ancsh_mlp_chain_grouped_fp_bf16x3 / _f16x2: kernel padded to 32 columns and split into the active
    scheme's planes, bias / scale / shift padded to 32 entries; cached on the layer dict (per scheme)"""
    from . import pointnet_util
    key = "split_head_" + pointnet_util.SPLIT_SCHEME
    if key not in layer:
        w = layer["w"]
        k, n = w.shape
        dev = w.device
        w32 = torch.zeros((k, 32), dtype=torch.float32, device=dev)
        w32[:, :n] = w
        pad = lambda v, fill: torch.cat([v, torch.full((32 - n,), fill, dtype=torch.float32, device=dev)]).contiguous()
        layer[key] = (pointnet_util._split_pack(w32), pad(layer["b"], 0.0), pad(layer["scale"], 1.0), pad(layer["shift"], 0.0))
    return layer[key]


def _tail_program(rows, K, mixed_pred, early_split_nocs, dev, bf16x3=False):
    """The one-tile chain program of ONE network (called inside its outer variable scope, the reference's 'SPFN'): fa_layer3's three
    convs, fc1 and every head, each layer rewriting the wave's tile in place.  The trunk `net` (fc1's output) has two 128-wide
    consumers only with early_split_nocs (fc11_1 and fc3_0, lib/architecture.py:111,198): fc1 then carries CH_SAVE and fc3_0
    CH_RESTORE.  -> (ops [5 ints per op], ptrs [5 per op], logits (rows, ld), ld, keep-alive list)."""
    out_dims, _ok = _head_dims(K, mixed_pred, early_split_nocs)
    with tf_util.variable_scope('est_net'):
        with tf_util.variable_scope('fa_layer3'):
            fp3 = [tf_util.get_layer(tf_util.current_scope('conv_%d' % i), dev) for i in range(3)]
        fc1 = tf_util.get_layer(tf_util.current_scope('fc1'), dev)
    n_head = sum(out_dims)
    ld = (n_head + 10 + 3) // 4 * 4
    logits = torch.empty((rows, ld), dtype=torch.float32, device=dev)
    ops, ptrs, keep = [], [], []

    def add(layer, act, flags=0, out_col=None):
        k, n = layer["w"].shape
        out = None if out_col is None else logits[:, out_col:]
        if bf16x3:
            # opt-in experiment (csrc/tail_bf16x3.hip): bf16x3-packed kernels; two register tiles, so no save / restore flags
            from . import pointnet_util
            par = _bf16x3_head_params(layer) if out is not None else (pointnet_util._bf16x3_weight(layer), layer["b"], layer["scale"], layer["shift"])
            ops.extend([k, n, 1 if act else 0, 0, ld if out is not None else 0])
            ptrs.extend([_lib.ptr(v) for v in par] + [_lib.ptr(out)])
            keep.append(layer)
            return
        ops.extend([k, n, 1 if act else 0, flags, ld if out is not None else 0])
        ptrs.extend([_lib.ptr(tf_util.packed_weight(layer)), _lib.ptr(layer["b"]), _lib.ptr(layer["scale"]), _lib.ptr(layer["shift"]), _lib.ptr(out)])
        keep.append(layer)

    add(fp3[0], True)
    add(fp3[1], True)
    add(fp3[2], True)
    add(fc1, True, CH_SAVE if early_split_nocs else 0)            # the tile = net (dropout = identity at test)
    with tf_util.variable_scope('nocs_net'):
        names = [tf_util.current_scope('fc2_{}'.format(i)) for i in range(len(out_dims))]
        if early_split_nocs:
            add(tf_util.get_layer(names[0], dev), False, 0, 0)                                   # W
            add(tf_util.get_layer_concat(names[2:], dev), False, 0, out_dims[0] + out_dims[1])   # scale | trans | confi
            add(tf_util.get_layer(tf_util.current_scope('fc11_1'), dev), False)                  # net -> fc11_1's output, in place
            add(tf_util.get_layer(names[1], dev), False, 0, out_dims[0])                         # nocs
        else:
            add(tf_util.get_layer_concat(names, dev), False, 0, 0)
    with tf_util.variable_scope('joint_net'):
        add(tf_util.get_layer(tf_util.current_scope('fc3_0'), dev), True, CH_RESTORE if early_split_nocs else 0)
        add(tf_util.get_layer(tf_util.current_scope('fc3_1'), dev), True)
        add(tf_util.get_layer_concat([tf_util.current_scope('fc4_{}'.format(i)) for i in range(4)], dev), False, 0, n_head)
    assert all(n == 128 or n <= 32 for n in ops[1::5])
    return ops, ptrs, logits, ld, keep


def run_tail_programs_bf16x3(programs, fp):
    """programs built with _tail_program(..., bf16x3=True); fp = (b, n, m, points2 (G * b, m, 128), idx, weight (b, n, 3), xyz (b, n, 3)), n % 64 == 0:
    ONE ancsh_mlp_chain_grouped_fp_bf16x3 launch per pair of networks (opt-in experiment, csrc/tail_bf16x3.hip)."""
    import ctypes
    b, n, m, points2, idx, weight, xyz = fp
    for g0 in range(0, len(programs), 2):
        grp = programs[g0:g0 + 2]
        k = len(grp)
        c_ops = [(ctypes.c_int * len(p[0]))(*p[0]) for p in grp]
        c_ptrs = [(ctypes.c_void_p * len(p[1]))(*p[1]) for p in grp]
        nops = (ctypes.c_int * k)(*[len(p[0]) // 5 for p in grp])
        ops_tab = (ctypes.c_void_p * k)(*[ctypes.cast(o, ctypes.c_void_p) for o in c_ops])
        ptr_tab = (ctypes.c_void_p * k)(*[ctypes.cast(o, ctypes.c_void_p) for o in c_ptrs])
        from . import pointnet_util
        _lib.call(pointnet_util.split_name("ancsh_mlp_chain_grouped_fp_bf16x3"), k, b, n, m, 128, _lib.ptr(points2[g0 * b:]), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(xyz),
                  ctypes.cast(nops, ctypes.c_void_p), ctypes.cast(ops_tab, ctypes.c_void_p), ctypes.cast(ptr_tab, ctypes.c_void_p))


def run_tail_programs(x, rows, programs, fp=None):
    """ONE ancsh_mlp_chain_grouped launch per pair of networks: x (G * rows, ldx) holds the networks' fa_layer3 input rows
    [interpolated (128) | xyz (3) | pad] network-major; programs[g] = _tail_program(...) of network g.
    fp = (b, n, m, points2 (G * b, m, 128), idx, weight (b, n, 3), xyz (b, n, 3)) instead of x: the rows are built in the chain's tile
    load (ancsh_mlp_chain_grouped_fp) -- no concat buffer, no interpolate + concat launch; same bits."""
    import ctypes
    G = len(programs)
    if fp is not None:
        b, n, m, points2, idx, weight, xyz = fp
        dev = points2.device
    else:
        dev = x.device
        ldx = x.shape[-1]
        x2 = x.reshape(G * rows, ldx)
    for g0 in range(0, G, 2):
        grp = programs[g0:g0 + 2]
        k = len(grp)
        need = any(f for p in grp for f in p[0][3::5])
        scratch = torch.empty((k * rows, 128), dtype=torch.float32, device=dev) if need else None
        c_ops = [(ctypes.c_int * len(p[0]))(*p[0]) for p in grp]
        c_ptrs = [(ctypes.c_void_p * len(p[1]))(*p[1]) for p in grp]
        nops = (ctypes.c_int * k)(*[len(p[0]) // 5 for p in grp])
        ops_tab = (ctypes.c_void_p * k)(*[ctypes.cast(o, ctypes.c_void_p) for o in c_ops])
        ptr_tab = (ctypes.c_void_p * k)(*[ctypes.cast(o, ctypes.c_void_p) for o in c_ptrs])
        if fp is not None:
            _lib.call("ancsh_mlp_chain_grouped_fp", k, b, n, m, 128, _lib.ptr(points2[g0 * b:]), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(xyz),
                      ctypes.cast(nops, ctypes.c_void_p), ctypes.cast(ops_tab, ctypes.c_void_p), ctypes.cast(ptr_tab, ctypes.c_void_p), _lib.ptr(scratch))
        else:
            _lib.call("ancsh_mlp_chain_grouped", k, rows, 131, _lib.ptr(x2[g0 * rows:]), ldx, ctypes.cast(nops, ctypes.c_void_p),
                      ctypes.cast(ops_tab, ctypes.c_void_p), ctypes.cast(ptr_tab, ctypes.c_void_p), _lib.ptr(scratch))


def _tail_chain(x, rows, K, mixed_pred, early_split_nocs):
    """fa_layer3's three convs, fc1 and every head of ONE network as one launch on x (rows, ld): the fa_layer3 input rows
    [interpolated (128) | xyz (3) | pad].  Called inside the network's outer variable scope (the reference's 'SPFN')."""
    prog = _tail_program(rows, K, mixed_pred, early_split_nocs, x.device)
    run_tail_programs(x, rows, [prog])
    return prog[2], prog[3]


def get_per_point_model_new(scope, P, n_max_parts, is_training, bn_decay, early_split=False, early_split_nocs=False,
                            mixed_pred=False, pred_joint=False, pred_joint_ind=False):
    '''
        Inputs:
            - P: BxNx3 tensor, the input point cloud
            - K := n_max_parts
        Outputs: a dict with W (BxNxK softmax), nocs_per_point (BxNx3K), confi_per_point (BxNx1),
            heatmap_per_point, unitvec_per_point, joint_axis_per_point, index_per_point and, when
            mixed_pred, gocs_per_point, global_scale, global_translation.
    '''
    K = n_max_parts
    _lib.require_cuda(P)
    P = P.contiguous().float()
    B, N, _ = P.shape
    rows = B * N
    fused = None
    if FUSED_TAIL:
        with tf_util.variable_scope(scope):
            fused = _fused_tail(scope, P, K, mixed_pred, early_split_nocs)
    if fused is not None:
        logits, ld = fused
        dev = P.device
    else:
      with tf_util.variable_scope(scope):
        out_dims = [K, 3 * K] + ([K, 3 * K] if mixed_pred else []) + [1]
        net = build_pointnet2_shared('est_net', X=P, out_dims=out_dims, is_training=is_training, bn_decay=bn_decay)
        dev = net.device
        n_head = sum(out_dims)
        ld = (n_head + 10 + 3) // 4 * 4
        logits = torch.empty((rows, ld), dtype=torch.float32, device=dev)

        with tf_util.variable_scope('nocs_net'):
            names = [tf_util.current_scope('fc2_{}'.format(i)) for i in range(len(out_dims))]
            if early_split_nocs:
                # columns of fc2_1 are produced from fc11_1's output below; keep a zero placeholder
                cat = tf_util.get_layer_concat([names[0]] + names[2:], dev, zero_cols={0: 3 * K})
            else:
                cat = tf_util.get_layer_concat(names, dev)
            tf_util.conv_rows(net, rows, 128, 128, cat, False, out=logits, ldy=ld)
            if early_split_nocs:
                l11 = tf_util.get_layer(tf_util.current_scope('fc11_1'), dev)
                shared = tf_util.conv_rows(net, rows, 128, 128, l11, False)     # no BN, no activation (:111)
                l21 = tf_util.get_layer(names[1], dev)
                tf_util.conv_rows(shared, rows, 128, 128, l21, False, out=logits[:, K:], ldy=ld)

        with tf_util.variable_scope('joint_net'):
            X = net
            for j in range(2):
                lay = tf_util.get_layer(tf_util.current_scope('fc3_{}'.format(j)), dev)
                X = tf_util.conv_rows(X, rows, 128, 128, lay, True)
            cat = tf_util.get_layer_concat([tf_util.current_scope('fc4_{}'.format(i)) for i in range(4)], dev)
            tf_util.conv_rows(X, rows, 128, 128, cat, False, out=logits[:, n_head:], ldy=ld)

    return _activations(logits, ld, B, N, K, mixed_pred)


def _activations(logits, ld, B, N, K, mixed_pred):
    """softmax / sigmoid / tanh of the head logits and the gocs composition (lib/architecture.py:124-159) in one launch."""
    dev = logits.device
    rows = B * N

    def new(c):
        return torch.empty((B, N, c), dtype=torch.float32, device=dev)

    pred = {
        'W': new(K), 'nocs_per_point': new(3 * K), 'confi_per_point': new(1), 'heatmap_per_point': new(1),
        'unitvec_per_point': new(3), 'joint_axis_per_point': new(3), 'index_per_point': new(3),
    }
    if mixed_pred:
        pred['gocs_per_point'] = new(3 * K)
        pred['global_scale'] = new(K)
        pred['global_translation'] = new(3 * K)
    _lib.call("ancsh_head_activations", rows, K, 1 if mixed_pred else 0, _lib.ptr(logits), ld,
              _lib.ptr(pred['W']), _lib.ptr(pred['nocs_per_point']), _lib.ptr(pred['confi_per_point']),
              _lib.ptr(pred['heatmap_per_point']), _lib.ptr(pred['unitvec_per_point']),
              _lib.ptr(pred['joint_axis_per_point']), _lib.ptr(pred['index_per_point']),
              _lib.ptr(pred.get('gocs_per_point')), _lib.ptr(pred.get('global_scale')),
              _lib.ptr(pred.get('global_translation')))
    return pred
