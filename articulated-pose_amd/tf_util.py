"""Inference-only counterparts of pointnet_plusplus/utils/tf_util.py (conv1d :52, conv2d :120,
batch_norm_template :512, dropout :594) on torch.Tensors resident on the MI355X.

The reference resolves parameters through TF variable scopes; the same mechanism is kept so call
sites read identically: `with variable_scope('layer1'): conv2d(x, 64, [1,1], scope='conv0', bn=True,...)`
looks up '<scopes>/layer1/conv0/{weights,biases,bn/*}' in the registered variable dict.
Every layer = ONE kernel launch (ancsh_conv1x1): f32-MFMA GEMM + bias + folded BN + ReLU.
"""
import contextlib

import torch

from . import _lib
from .weights import fold_layer

relu = "relu"   # stand-in for tf.nn.relu as an `activation_fn` value

_state = {"weights": None, "cache": {}, "scopes": []}
_caches = {}   # id(weights dict) -> (weights, folded device tensors): several networks stay resident


def set_variables(weights):
    """Make a {tf variable name: ndarray} dict (see weights.py) the current variable store.  Folded
    device copies are cached per store, so switching between networks costs nothing."""
    _state["weights"] = weights
    ent = _caches.get(id(weights))
    if ent is None or ent[0] is not weights:
        ent = (weights, {})
        _caches[id(weights)] = ent
    _state["cache"] = ent[1]


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    _state["scopes"].append(name)
    try:
        yield "/".join(_state["scopes"])
    finally:
        _state["scopes"].pop()


def current_scope(*more):
    return "/".join(list(_state["scopes"]) + [m for m in more if m])


def get_layer(full_scope, device):
    """Folded device tensors (w, b, scale, shift) of one conv layer, cached per variable store."""
    key = (full_scope, str(device))
    hit = _state["cache"].get(key)
    if hit is None:
        if _state["weights"] is None:
            raise RuntimeError("tf_util.set_variables(weights) must be called before building the model")
        folded = fold_layer(_state["weights"], full_scope)
        hit = {k: torch.from_numpy(v).to(device) for k, v in folded.items()}
        _state["cache"][key] = hit
    return hit


PACKED_CONV = __import__('os').environ.get('ANCSH_PACKED_CONV', '0') != '0'   # opt-in: wide layers through csrc/conv_packed.hip (same bits; pays off from ~64k rows or k >= 1024)


def use_packed(rows, cin, cout, ldx, x, pool=0):
    """Route a layer to ancsh_conv1x1_packed?  Same bits either way; the packed entry holds (i) the small-layer schedule for the
    backbone's 128 / 256 / 259 / 384-channel layers without pooling (csrc/conv_rowtile.hip: 8-13 us instead of 12-17 per launch
    at 4096..16384 rows), (ii) the wave-independent kernel, which wins on the widest pooled layer (4096 x 512 -> 1024: 42 vs
    48 us); ANCSH_PACKED_CONV=1 sends every eligible wide layer there."""
    if pool == 0 and cin in (128, 256, 259, 384) and cout % 128 == 0:
        return True
    aligned = ldx % 4 == 0 and x.data_ptr() % 16 == 0
    if aligned and cout % 64 == 0 and rows >= 1024 and (PACKED_CONV or (pool != 0 and cout >= 1024)):
        return True
    return False


def packed_weight(layer, row0=0):
    """The layer's kernel rows [row0:] in the MFMA fragment order of ancsh_sa_pack_weights, cached on the layer dict."""
    key = "w_packed" if row0 == 0 else "w_packed_from_%d" % row0
    if key not in layer:
        from . import _lib
        w = layer["w"][row0:]
        k, n = w.shape
        packed = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(k, n), dtype=torch.float32, device=w.device)
        _lib.call("ancsh_sa_pack_weights", k, n, _lib.ptr(w), _lib.ptr(packed))
        layer[key] = packed
    return layer[key]


def sa_first_layer_split(layer):
    """First layer of a set-abstraction level WITH input features, whose input row is [x_j - c | f_j] (pointnet_util.py:55): its
    dot product is summed features first, coordinates last (see include/ancsh_hip.h, ancsh_sa_module_fused_partial).  Cached on
    the layer dict:
      "w_feat"        kernel rows 3.. (a view: the rows are contiguous) -- the per-point partial sums
      "w_xyz_packed"  kernel rows 0..2 in ancsh_sa_pack_weights order   -- the fused kernel's continuation
      "w_feat_first"  kernel rows re-ordered [3.., 0..2]                -- the unfused path runs the same chain on [f_j | x_j - c]"""
    if "w_feat" not in layer:
        from . import _lib
        w = layer["w"]
        layer["w_feat"] = w[3:]
        wx = w[:3].contiguous()
        packed = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(3, w.shape[1]), dtype=torch.float32, device=w.device)
        _lib.call("ancsh_sa_pack_weights", 3, w.shape[1], _lib.ptr(wx), _lib.ptr(packed))
        layer["w_xyz_packed"] = packed
        layer["w_feat_first"] = torch.cat([w[3:], w[:3]], dim=0).contiguous()
    return layer


def get_layer_sa_packed(full_scope, device):
    """get_layer plus "w_packed": the kernel in ancsh_sa_module_fused's fragment order, cached."""
    layer = get_layer(full_scope, device)
    packed_weight(layer)
    return layer


def get_layer_concat(full_scopes, device, zero_cols=()):
    """Column-wise concatenation of several layers that share their input (each output column is an
    independent dot product, so concatenating kernels is exact).  zero_cols: extra zero columns to
    append after layer i, as {i: n}."""
    key = ("cat", tuple(full_scopes), tuple(sorted(dict(zero_cols).items())), str(device))
    hit = _state["cache"].get(key)
    if hit is None:
        parts = {k: [] for k in ("w", "b", "scale", "shift")}
        zc = dict(zero_cols)
        for i, s in enumerate(full_scopes):
            f = fold_layer(_state["weights"], s)
            for k in parts:
                parts[k].append(torch.from_numpy(f[k]))
            if i in zc:
                cin = f["w"].shape[0]
                parts["w"].append(torch.zeros(cin, zc[i]))
                parts["b"].append(torch.zeros(zc[i]))
                parts["scale"].append(torch.ones(zc[i]))
                parts["shift"].append(torch.zeros(zc[i]))
        hit = {k: torch.cat(v, dim=-1).contiguous().to(device) for k, v in parts.items()}
        _state["cache"][key] = hit
    return hit


def conv_rows(x, rows, cin, ldx, layer, act, out=None, ldy=None, pool=0):
    """y[rows(/pool), cout] = act(BN(x[rows,cin] @ w + b)) -- one ancsh_conv1x1 launch.
    x: tensor whose storage holds `rows` rows of stride `ldx`; out: optional destination whose
    first element is y[0,0] with row stride ldy."""
    cout = layer["w"].shape[1]
    orows = rows // pool if pool else rows
    if out is None:
        out = torch.empty((orows, cout), dtype=torch.float32, device=x.device)
        ldy = cout
    if use_packed(rows, cin, cout, ldx, x, pool):
        # pre-packed weights: the small-layer schedule (csrc/conv_rowtile.hip) or the wave-independent kernel (csrc/conv_packed.hip)
        _lib.call("ancsh_conv1x1_packed", rows, cin, cout, _lib.ptr(x), ldx, _lib.ptr(packed_weight(layer)), _lib.ptr(layer["b"]),
                  _lib.ptr(layer["scale"]), _lib.ptr(layer["shift"]), 1 if act else 0, _lib.ptr(out), ldy, pool, None, 0)
        return out
    _lib.call("ancsh_conv1x1", rows, cin, cout, _lib.ptr(x), ldx, _lib.ptr(layer["w"]), _lib.ptr(layer["b"]),
              _lib.ptr(layer["scale"]), _lib.ptr(layer["shift"]), 1 if act else 0, _lib.ptr(out), ldy, pool)
    return out


def _rows_view(inputs):
    """(rows, cin, ldx) of a tensor whose last dim is the channel and whose leading dims are dense."""
    cin = inputs.shape[-1]
    if inputs.stride(-1) != 1 and cin > 1:
        inputs = inputs.contiguous()
    ldx = inputs.stride(-2) if inputs.dim() >= 2 else cin
    rows = inputs.numel() // cin if cin else 0
    # leading dims must be dense w.r.t. the row stride
    exp = ldx
    for d in range(inputs.dim() - 2, -1, -1):
        if inputs.shape[d] != 1 and inputs.stride(d) != exp:
            inputs = inputs.contiguous()
            return inputs, rows, cin, cin
        exp *= inputs.shape[d]
    return inputs, rows, cin, ldx


def _conv(inputs, num_output_channels, scope, bn, activation_fn, is_training):
    if is_training not in (False, None):
        raise ValueError("inference-only build: is_training must be False (training is out of scope)")
    _lib.require_cuda(inputs)
    inputs, rows, cin, ldx = _rows_view(inputs.float())
    layer = get_layer(current_scope(scope), inputs.device)
    if layer["w"].shape != (cin, num_output_channels):
        raise ValueError(f"{current_scope(scope)}: kernel {tuple(layer['w'].shape)} does not match "
                         f"input channels {cin} -> {num_output_channels}")
    y = conv_rows(inputs, rows, cin, ldx, layer, activation_fn is not None)
    return y.view(*inputs.shape[:-1], num_output_channels)


def conv1d(inputs, num_output_channels, kernel_size, scope, stride=1, padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=relu, bn=False, bn_decay=None,
           is_training=None):
    """1x1 'conv1d' on BxLxC (tf_util.py:52-117): conv -> bias -> [BN] -> [activation]."""
    if kernel_size != 1 or stride != 1 or data_format != 'NHWC':
        raise NotImplementedError("only kernel_size=1, stride=1, NHWC is on the ANCSH graph")
    return _conv(inputs, num_output_channels, scope, bn, activation_fn, is_training)


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=relu, bn=False, bn_decay=None,
           is_training=None):
    """1x1 'conv2d' on BxHxWxC (tf_util.py:120-185)."""
    if list(kernel_size) != [1, 1] or list(stride) != [1, 1] or data_format != 'NHWC':
        raise NotImplementedError("only 1x1 kernels, stride 1, NHWC are on the ANCSH graph")
    return _conv(inputs, num_output_channels, scope, bn, activation_fn, is_training)


def dropout(inputs, is_training, scope, keep_prob=0.5, noise_shape=None):
    """tf_util.py:594-615: identity at inference."""
    if is_training not in (False, None):
        raise ValueError("inference-only build: is_training must be False")
    return inputs
