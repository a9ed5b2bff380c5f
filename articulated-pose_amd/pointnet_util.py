"""Inference-only counterparts of pointnet_plusplus/utils/pointnet_util.py: sample_and_group (:29),
sample_and_group_all (:66), pointnet_sa_module (:94), pointnet_fp_module (:206) -- same names,
argument order and return tuples, on torch.Tensors resident on the MI355X.

Kernel-level fusions relative to the reference graph (results unchanged):
  * farthest_point_sample + gather_point -> one launch (new_xyz written by the FPS kernel);
  * group_point(xyz) - new_xyz, group_point(points) and the concat -> written straight into one
    (B, npoint, nsample, 3+C) buffer by two gather launches (no tile/sub/concat passes);
  * every conv2d+bias+BN+ReLU = one MFMA launch; the last SA layer also folds tf.reduce_max;
  * three_nn -> weights -> three_interpolate -> concat: weights in one launch, interpolation writes
    directly into the concat buffer.
"""
import torch

from . import _lib, tf_util
from .tf_ops import tf_grouping, tf_interpolate, tf_sampling
from .tf_ops.tf_sampling import farthest_point_sample, gather_point      # noqa: F401  (re-exported like the reference)
from .tf_ops.tf_grouping import query_ball_point, group_point, knn_point  # noqa: F401
from .tf_ops.tf_interpolate import three_nn, three_interpolate           # noqa: F401


def fps_sampling(npoint, xyz):
    return farthest_point_sample(npoint, xyz)


def gather_nd_point(P, sample_index):
    return gather_point(P, sample_index)


def _pad4(c):
    return (c + 3) // 4 * 4


class Geometry(dict):
    """Weight-independent results of one forward (FPS picks, ball-query indices, 3-NN indices + weights), keyed by
    layer scope.  The ANCSH and the NPCS network see the same cloud, so the pipeline computes these once per batch
    and hands them to the second network (`Network.predict(P, geometry=...)`); values are what the op kernels
    returned, bit for bit."""


_geom = {"cur": None}


def use_geometry(g):
    """Install a Geometry to record into / replay from for the model built next (None = off)."""
    _geom["cur"] = g


def _geom_get(key):
    g = _geom["cur"]
    return None if g is None else g.get(key)


def _geom_put(key, val):
    if _geom["cur"] is not None:
        _geom["cur"][key] = val


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True):
    '''FPS centroids -> ball query -> gathered neighbourhoods (pointnet_util.py:29-64).
    xyz (B, n, 3), points (B, n, C) or None.  Returns new_xyz (B, npoint, 3); new_points (B, npoint, nsample, 3 + C) whose first
    three channels are the neighbours' coordinates relative to their centroid (just those when points is None, features only
    when use_xyz is False); idx (B, npoint, nsample) int32; grouped_xyz (B, npoint, nsample, 3), centred.
    knn=True groups the nsample nearest points instead of the ball query (off the ANCSH graph).'''
    xyz = xyz.contiguous().float()
    b, n, _ = xyz.shape
    if knn:                                           # pointnet_util.py:49-50 (never taken by the ANCSH graph: knn=False everywhere)
        new_xyz = tf_sampling.farthest_point_sample_gather(npoint, xyz)[1]
        idx = tf_grouping.knn_point(nsample, xyz, new_xyz)[1]
    else:
        new_xyz, idx = _sample_and_query(npoint, radius, nsample, xyz)
    c = 0 if points is None else points.shape[2]
    use_feat = points is not None and c > 0
    width = (3 if (use_xyz or not use_feat) else 0) + (c if use_feat else 0)
    ld = _pad4(width)
    buf = torch.empty((b, npoint, nsample, ld), dtype=torch.float32, device=xyz.device)
    if ld != width:
        buf[..., width:].zero_()
    off = 0
    if use_xyz or not use_feat:
        _lib.call("ancsh_group_point_ex", b, n, 3, npoint, nsample, _lib.ptr(xyz), _lib.ptr(idx), _lib.ptr(new_xyz),
                  _lib.ptr(buf), ld, 0)
        off = 3
    if use_feat:
        points = points.contiguous().float()
        _lib.call("ancsh_group_point_ex", b, n, c, npoint, nsample, _lib.ptr(points), _lib.ptr(idx), 0,
                  _lib.ptr(buf), ld, off)
    new_points = buf[..., :width]
    grouped_xyz = buf[..., :3] if (use_xyz or not use_feat) else None
    return new_xyz, new_points, idx, grouped_xyz


_GROUP_ALL_CONST = {}


def sample_and_group_all(xyz, points, use_xyz=True):
    '''The whole cloud as ONE neighbourhood around the origin (pointnet_util.py:66-91): new_xyz (B, 1, 3) zeros,
    new_points (B, 1, n, 3 + C) = [xyz | points], idx (B, 1, n) = 0..n-1, grouped_xyz (B, 1, n, 3).'''
    b, n, _ = xyz.shape
    key = (b, n, str(xyz.device))
    if key not in _GROUP_ALL_CONST:          # constants: built once (outside any graph capture), not three launches per call
        _GROUP_ALL_CONST[key] = (torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device),
                                 torch.arange(n, dtype=torch.int32, device=xyz.device).view(1, 1, n).repeat(b, 1, 1))
    new_xyz, idx = _GROUP_ALL_CONST[key]
    grouped_xyz = xyz.reshape(b, 1, n, 3)
    if points is not None:
        new_points = torch.cat([xyz, points], dim=2) if use_xyz else points
        new_points = new_points.unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


import os as _os

FP_SINGLE_SOURCE = _os.environ.get('ANCSH_FP_SINGLE_SOURCE', '1') != '0'   # exact shortcut for a one-point interpolation source
FUSED_SA = _os.environ.get('ANCSH_FUSED_SA', '1') != '0'     # one-launch SA body (csrc/sa_fused.hip); False = op-by-op path (same results, bit for bit)
_FUSED_SHAPES = {(0, (64, 64, 128)), (128, (128, 128, 256))}
# EXPERIMENT (opt-in, see csrc/sa_bf16x3.hip): the fused SA levels on the bf16 matrix pipe with f32 products emulated by six bf16
# products.  Not bit-identical to the f32 path (the additions inside the instruction are ordered differently), hence off by default.
SA_BF16X3 = int(_os.environ.get('ANCSH_SA_BF16X3', '0'))      # 1: the level without input features (register-resident kernel); 2: both levels;
#                                                             3: + the tail chain (paired forward only: csrc/tail_bf16x3.hip); 4: + the mid-section (csrc/mid_bf16x3.hip)


# the split scheme of the experiment (csrc/bx3.h): 'bf16x3' = three bf16 terms, six products (f32-exact products); 'f16x2' = two f16 terms
# (hi + 2^-11 mid), three products into two accumulators (~22 bits per operand, f16's range)
SPLIT_SCHEME = _os.environ.get('ANCSH_SPLIT_SCHEME', 'bf16x3')


def split_name(entry):
    """ABI name of a split-16 entry point for the active scheme: '..._bf16x3...' -> '..._f16x2...' when SPLIT_SCHEME == 'f16x2'"""
    if SPLIT_SCHEME not in ('bf16x3', 'f16x2'):
        raise ValueError("ANCSH_SPLIT_SCHEME must be bf16x3 or f16x2, got %r" % (SPLIT_SCHEME,))
    return entry.replace('bf16x3', SPLIT_SCHEME)


def _split_pack(w):
    """(k, n % 32 == 0) f32 kernel on the device -> its planes in MFMA fragment order for the active scheme"""
    k, n = w.shape
    nbytes = getattr(_lib.lib(), split_name("ancsh_sa_packed_weight_bytes_bf16x3"))(k, n)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.call(split_name("ancsh_sa_pack_weights_bf16x3"), k, n, _lib.ptr(w), _lib.ptr(packed))
    return packed


def _bf16x3_weight(layer, row0=0):
    """the layer's kernel (rows row0..) split into the scheme's 16-bit planes in MFMA fragment order, cached on the layer dict (per scheme)"""
    key = "w_%s_%d" % (SPLIT_SCHEME, row0)
    if key not in layer:
        layer[key] = _split_pack(layer["w"][row0:].contiguous())
    return layer[key]


def _bf16x3_xyz_weight(first, c1):
    """the three coordinate rows of a feature level's first layer (tf_util.sa_first_layer_split) in the split packing, cached"""
    key = "w_xyz_" + SPLIT_SCHEME
    if key not in first:
        first[key] = _split_pack(first["w"][:3].contiguous())
    return first[key]


def _sample_and_query(npoint, radius, nsample, xyz):
    key = ("sa", tf_util.current_scope(), npoint, float(radius), nsample)
    hit = _geom_get(key)
    if hit is None:
        _, new_xyz = tf_sampling.farthest_point_sample_gather(npoint, xyz)
        idx, _cnt = tf_grouping.query_ball_point(radius, nsample, xyz, new_xyz)
        hit = (new_xyz, idx)
        _geom_put(key, hit)
    return hit


def precompute_geometry(xyz, geometry, scope='SPFN/est_net'):
    """Fill `geometry` with every weight-independent result of the backbone for the cloud batch xyz (B, N, 3) -- the FPS picks
    and ball-query indices of layer1 / layer2, the 3-NN indices + weights of fa_layer2 / fa_layer3 (fa_layer1 interpolates from
    a single point and needs none) -- under the keys the modules look up, so that networks built afterwards (on any stream
    ordered after this call) replay them instead of computing them.  Same kernels, same values as the lazy path."""
    from .architectures import SA_LEVELS
    xyz = xyz.contiguous().float()
    levels = [xyz]
    for name, npoint, radius, nsample, _mlp, group_all in SA_LEVELS:
        if group_all:
            break
        key = ("sa", scope + "/" + name, npoint, float(radius), nsample)
        _, new_xyz = tf_sampling.farthest_point_sample_gather(npoint, levels[-1])
        idx, _cnt = tf_grouping.query_ball_point(radius, nsample, levels[-1], new_xyz)
        geometry[key] = (new_xyz, idx)
        levels.append(new_xyz)
    for name, fine in (("fa_layer2", 1), ("fa_layer3", 0)):
        _dist, idx, weight = tf_interpolate.three_nn_weights(levels[fine], levels[fine + 1])
        geometry[("fp", scope + "/" + name)] = (idx, weight)
    return geometry


def _try_fused_sa(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, knn, use_xyz):
    import ctypes
    if not FUSED_SA or group_all or knn or mlp2 is not None or not use_xyz or nsample != 64:
        return None
    c = 0 if points is None else points.shape[2]
    b, n, _ = xyz.shape
    if (c, tuple(mlp)) not in _FUSED_SHAPES:
        return None
    xyz = xyz.contiguous().float()
    new_xyz, idx = _sample_and_query(npoint, radius, nsample, xyz)
    layers = [tf_util.get_layer_sa_packed(tf_util.current_scope('conv%d' % i), xyz.device) for i in range(3)]
    out = torch.empty((b, npoint, mlp[2]), dtype=torch.float32, device=xyz.device)
    if SA_BF16X3 >= 1 and c == 0:
        ptrs = (ctypes.c_void_p * 12)(*[_lib.ptr(v) for l in layers for v in (_bf16x3_weight(l), l["b"], l["scale"], l["shift"])])
        _lib.call(split_name("ancsh_sa_module_fused_bf16x3_grouped"), 1, b, n, npoint, nsample, 0, mlp[0], mlp[1], mlp[2], _lib.ptr(xyz), None,
                  _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
        return new_xyz, out, idx
    if c == 0:
        ptrs = (ctypes.c_void_p * 12)(*[_lib.ptr(l[k]) for l in layers for k in ("w_packed", "b", "scale", "shift")])
        _lib.call("ancsh_sa_module_fused", b, n, npoint, nsample, 0, mlp[0], mlp[1], mlp[2], _lib.ptr(xyz), None,
                  _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
        return new_xyz, out, idx
    # The first layer sums the feature channels first and the centred coordinates last: the feature part of every neighbour's
    # dot product is one per-POINT partial sum (n rows instead of 64 * npoint), gathered by the fused kernel like a feature row
    first = tf_util.sa_first_layer_split(layers[0])
    feats = points.contiguous().float()
    partial = torch.empty((b, n, mlp[0]), dtype=torch.float32, device=xyz.device)
    if tf_util.use_packed(b * n, c, mlp[0], c, feats):
        _lib.call("ancsh_conv1x1_packed", b * n, c, mlp[0], _lib.ptr(feats), c, _lib.ptr(tf_util.packed_weight(layers[0], 3)), None, None,
                  None, 2, _lib.ptr(partial), mlp[0], 0, None, 0)
    else:
        _lib.call("ancsh_conv1x1", b * n, c, mlp[0], _lib.ptr(feats), c, _lib.ptr(first["w_feat"]), None, None, None, 2, _lib.ptr(partial),
                  mlp[0], 0)
    if SA_BF16X3 >= 2:
        ptrs = (ctypes.c_void_p * 12)(*([_lib.ptr(_bf16x3_xyz_weight(first, mlp[0]))] + [_lib.ptr(first[k]) for k in ("b", "scale", "shift")] +
                                        [_lib.ptr(v) for l in layers[1:] for v in (_bf16x3_weight(l), l["b"], l["scale"], l["shift"])]))
        _lib.call(split_name("ancsh_sa_module_fused_partial_bf16x3_grouped"), 1, b, n, npoint, nsample, mlp[0], mlp[1], mlp[2], _lib.ptr(xyz), _lib.ptr(partial),
                  _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
        return new_xyz, out, idx
    ptrs = (ctypes.c_void_p * 12)(*([_lib.ptr(first["w_xyz_packed"])] + [_lib.ptr(first[k]) for k in ("b", "scale", "shift")] +
                                    [_lib.ptr(l[k]) for l in layers[1:] for k in ("w_packed", "b", "scale", "shift")]))
    _lib.call("ancsh_sa_module_fused_partial", b, n, npoint, nsample, mlp[0], mlp[1], mlp[2], _lib.ptr(xyz), _lib.ptr(partial),
              _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
    return new_xyz, out, idx


def pointnet_sa_module(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, is_training, bn_decay, scope,
                       bn=True, pooling='max', knn=False, use_xyz=True, use_nchw=False, reuse=False):
    '''Set-abstraction level (pointnet_util.py:94-161): sample + group, the shared MLP `mlp` on every neighbour, max over the
    neighbourhood, optional `mlp2` on the pooled vector.  Returns new_xyz (B, npoint, 3), new_points (B, npoint, mlp[-1] or
    mlp2[-1]) and idx (B, npoint, nsample) int32.  Only pooling='max', NHWC, inference.'''
    if pooling != 'max':
        raise NotImplementedError("only pooling='max' is on the ANCSH graph")
    if use_nchw:
        raise NotImplementedError("NHWC only")
    with tf_util.variable_scope(scope):
        fused = _try_fused_sa(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, knn, use_xyz)
        if fused is not None:
            return fused
        if group_all:
            new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, use_xyz)
        else:
            new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, nsample, xyz, points, knn, use_xyz)
        b, m, ns, cin = new_points.shape
        rows = b * m * ns
        x, ldx = new_points, new_points.stride(2)
        if new_points.stride(3) != 1 or new_points.stride(1) != ns * ldx or (b > 1 and new_points.stride(0) != m * ns * ldx):
            x = new_points.contiguous()
            ldx = cin
        fuse_pool = mlp2 is None and ns in (64, 128)
        feat_first = (not group_all) and use_xyz and points is not None and points.shape[2] > 0
        if feat_first:
            # same summation order as the fused kernel and the oracle: feature channels first, centred coordinates last -- gathered
            # straight into that layout (features at column 0, x - c behind them) instead of re-concatenating the grouped tensor
            c = points.shape[2]
            ld = _pad4(cin)
            x = torch.empty((b, m, ns, ld), dtype=torch.float32, device=xyz.device)
            if ld != cin:
                x[..., cin:].zero_()
            xyz_c, feats_c = xyz.contiguous().float(), points.contiguous().float()
            _lib.call("ancsh_group_point_ex", b, xyz.shape[1], c, m, ns, _lib.ptr(feats_c), _lib.ptr(idx), 0, _lib.ptr(x), ld, 0)
            _lib.call("ancsh_group_point_ex", b, xyz.shape[1], 3, m, ns, _lib.ptr(xyz_c), _lib.ptr(idx), _lib.ptr(new_xyz), _lib.ptr(x), ld, c)
            ldx = ld
        for i, num_out_channel in enumerate(mlp):
            layer = tf_util.get_layer(tf_util.current_scope('conv%d' % i), x.device)
            if i == 0 and feat_first:
                # the re-ordered first layer lives on the cached layer dict, so its packed weights are cached with it
                split = tf_util.sa_first_layer_split(layer)
                if "_feat_first_layer" not in split:
                    split["_feat_first_layer"] = dict(w=split["w_feat_first"], b=layer["b"], scale=layer["scale"], shift=layer["shift"])
                layer = split["_feat_first_layer"]
            last = i == len(mlp) - 1
            x = tf_util.conv_rows(x, rows, cin, ldx, layer, True, pool=ns if (last and fuse_pool) else 0)
            cin = ldx = num_out_channel
        if fuse_pool:
            new_points = x.view(b, m, cin)
        else:
            y = torch.empty((b * m, cin), dtype=torch.float32, device=x.device)
            _lib.call("ancsh_group_max", b * m, ns, cin, _lib.ptr(x), _lib.ptr(y))
            new_points = y.view(b, m, 1, cin)
            if mlp2 is not None:
                for i, num_out_channel in enumerate(mlp2):
                    new_points = tf_util.conv2d(new_points, num_out_channel, [1, 1], padding='VALID', stride=[1, 1],
                                                bn=bn, is_training=is_training, scope='conv_post_%d' % i, bn_decay=bn_decay)
            new_points = new_points.squeeze(2)
        return new_xyz, new_points, idx


def pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=True):
    '''Feature-propagation level (pointnet_util.py:206-236): features points2 (B, n2, C2) of the sparser level xyz2 are
    interpolated to the n1 points of xyz1 (three nearest neighbours, inverse squared-distance weights), concatenated in front
    of the skip features points1 (B, n1, C1) and pushed through the shared MLP `mlp`.  Returns (B, n1, mlp[-1]).'''
    with tf_util.variable_scope(scope):
        if FP_SINGLE_SOURCE and xyz2.shape[1] == 1 and points1 is not None and len(mlp) >= 1:
            return _fp_single_source(points1, points2, mlp)
        buf = fp_interpolate_concat(xyz1, xyz2, points1, points2)
        b, n, ld = buf.shape
        width = points2.shape[2] + (0 if points1 is None else points1.shape[2])
        x, rows, cin, ldx = buf, b * n, width, ld
        for i, num_out_channel in enumerate(mlp):
            layer = tf_util.get_layer(tf_util.current_scope('conv_%d' % i), x.device)
            x = tf_util.conv_rows(x, rows, cin, ldx, layer, True)
            cin = ldx = num_out_channel
        return x.view(b, n, cin)


def _fp_single_source(points1, points2, mlp):
    """FP module whose sparser level is ONE point per cloud (the ANCSH fa_layer1 under a group_all SA level).
    three_nn then returns that point three times with distances (d, inf, inf), the weights are exactly (1, 0, 0)
    ((1/d)/(1/d + 0 + 0)), and the interpolated block of the concat [interpolated | points1] is the same vector g for
    every point of the cloud (pointnet_util.py:218-229).  The first conv's k-ordered dot product therefore starts with
    the same partial sum for all of a cloud's points: compute it once per cloud (raw accumulators) and let the big conv
    continue the chain from it over the points1 channels only -- bit-identical to the materialised path, without the
    (B, n, c2 + c1) buffer and 5x fewer flops for the 1024 + 256 -> 256 layer."""
    b, n, c1 = points1.shape
    c2 = points2.shape[2]
    dev = points1.device
    layer = tf_util.get_layer(tf_util.current_scope('conv_0'), dev)
    cout = layer["w"].shape[1]
    g = points2.reshape(b, c2).contiguous().float()
    init = torch.empty((b, cout), dtype=torch.float32, device=dev)
    _lib.call("ancsh_conv1x1", b, c2, cout, _lib.ptr(g), c2, _lib.ptr(layer["w"]), None, None, None, 2, _lib.ptr(init), cout, 0)
    p1 = points1.contiguous().float()
    x = torch.empty((b * n, cout), dtype=torch.float32, device=dev)
    if tf_util.use_packed(b * n, c1, cout, c1, p1):
        _lib.call("ancsh_conv1x1_packed", b * n, c1, cout, _lib.ptr(p1), c1, _lib.ptr(tf_util.packed_weight(layer, c2)),
                  _lib.ptr(layer["b"]), _lib.ptr(layer["scale"]), _lib.ptr(layer["shift"]), 1, _lib.ptr(x), cout, 0, _lib.ptr(init), n)
    else:
        _lib.call("ancsh_conv1x1_ex", b * n, c1, cout, _lib.ptr(p1), c1, _lib.ptr(layer["w"][c2:]), _lib.ptr(layer["b"]),
                  _lib.ptr(layer["scale"]), _lib.ptr(layer["shift"]), 1, _lib.ptr(x), cout, 0, _lib.ptr(init), n)
    cin = cout
    for i, num_out_channel in enumerate(mlp[1:], start=1):
        layer = tf_util.get_layer(tf_util.current_scope('conv_%d' % i), dev)
        x = tf_util.conv_rows(x, b * n, cin, cin, layer, True)
        cin = num_out_channel
    return x.view(b, n, cin)


def fp_interpolate_concat(xyz1, xyz2, points1, points2):
    """three_nn -> inverse-distance weights -> three_interpolate -> concat [interpolated, points1]
    (pointnet_util.py:218-229) into one (B, n, ld) buffer, ld = width padded to a multiple of 4 (zero pad).
    Must be called inside the FP module's variable scope (the 3-NN results are cached per scope)."""
    key = ("fp", tf_util.current_scope())
    hit = _geom_get(key)
    if hit is None:
        _dist, idx, weight = tf_interpolate.three_nn_weights(xyz1, xyz2)      # max(dist,1e-10); (1/dist)/sum(1/dist), same launch
        _geom_put(key, (idx, weight))
    else:
        idx, weight = hit
    b, n, _ = xyz1.shape
    m, c2 = points2.shape[1], points2.shape[2]
    c1 = 0 if points1 is None else points1.shape[2]
    width = c2 + c1
    ld = _pad4(width)
    buf = torch.empty((b, n, ld), dtype=torch.float32, device=xyz1.device)
    points2 = points2.contiguous().float()
    if c2 % 4 == 0:
        # interpolation, the points1 block and the zero pad in ONE launch (was three: interpolate + slice copy + fill)
        p1 = None if points1 is None else points1.contiguous().float()
        _lib.call("ancsh_fp_interpolate_concat", b, m, c2, n, _lib.ptr(points2), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(p1), c1,
                  _lib.ptr(buf), ld)
        return buf
    _lib.call("ancsh_three_interpolate_ex", b, m, c2, n, _lib.ptr(points2), _lib.ptr(idx), _lib.ptr(weight),
              _lib.ptr(buf), ld, 0)
    if c1:
        buf[..., c2:width] = points1        # concat [interpolated, points1] (:226)
    if ld != width:
        buf[..., width:].zero_()
    return buf
