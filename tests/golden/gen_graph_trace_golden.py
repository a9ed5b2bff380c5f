"""Generator of tests/golden/graph_trace.json -- BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

What it pins: the WIRING of the inference graph, taken from the reference's own graph-building files executed unmodified:
    pointnet_plusplus/utils/tf_util.py          (conv1d / conv2d / batch_norm_for_* / dropout)
    pointnet_plusplus/utils/pointnet_util.py    (sample_and_group[_all], pointnet_sa_module, pointnet_fp_module)
    pointnet_plusplus/architectures.py:56-95    (build_pointnet2_shared)
    lib/architecture.py:86-161,195-208          (get_per_point_model_new, joint_est_model)
called the way lib/network.py:57-68 calls them (scope 'SPFN', flags of main.py:42-52 for --nocs_type ancsh / npcs).

What it does NOT pin: arithmetic.  TensorFlow is absent from this image and from /root/reference; the `tensorflow` module these
files import here is a RECORDER that computes nothing: a tensor is a (id, static shape) pair and every op appends one record
{op, scope, inputs (ids, in operand order), output id, shape, attributes} to the trace in call order.  The native-op wrapper modules
(tf_sampling / tf_grouping / tf_interpolate: tf.load_op_library shells) are recorders too.  Two things in the recorder are knowledge
about TensorFlow 1.x rather than something read from the reference: tf.contrib.layers.batch_norm(center=True, scale=True) owns the
variables beta, gamma, moving_mean, moving_variance under its scope, and tf.cond(False, a, b) evaluates b (is_training = False).

The trace removes one failure mode the numeric tests cannot see: oracle/net_oracle.py and the product sharing the SAME wiring mistake
(layer order, concat operand order, BN placement, activation, pooling axis, scope names).  tests/test_graph_trace_cpu.py compares it
with the product's level tables / variable inventory and with the oracle (whose concat orders are probed numerically).

    python tests/golden/gen_graph_trace_golden.py          # rewrites tests/golden/graph_trace.json
"""
import contextlib
import json
import os
import sys
import types

REF = os.environ.get("ANCSH_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


class Dim(object):
    def __init__(self, v):
        self.value = v

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dim) else o)

    def __hash__(self):
        return hash(self.value)


class Shape(object):
    def __init__(self, dims):
        self.dims = list(dims)

    def __getitem__(self, i):
        return Dim(self.dims[i])

    def as_list(self):
        return list(self.dims)

    def __len__(self):
        return len(self.dims)


class Recorder(object):
    def __init__(self):
        self.records, self.variables, self.scopes, self.n = [], [], [], 0

    def scope(self):
        return "/".join(self.scopes)

    def tensor(self, shape, op, inputs=(), **attrs):
        self.n += 1
        t = Tensor(self, self.n, shape)
        rec = {"op": op, "scope": self.scope(), "in": [x.id if isinstance(x, Tensor) else x for x in inputs], "out": t.id,
               "shape": [d if isinstance(d, int) else None for d in shape]}
        rec.update({k: v for k, v in attrs.items() if v is not None})
        self.records.append(rec)
        return t


class Tensor(object):
    def __init__(self, rec, id_, shape):
        self.rec, self.id, self.dims = rec, id_, list(shape)

    def get_shape(self):
        return Shape(self.dims)

    @property
    def shape(self):
        return Shape(self.dims)

    def _bin(self, op, other, swap=False):
        a, b = (other, self) if swap else (self, other)
        ins = [x if isinstance(x, Tensor) else "const:%r" % (x,) for x in (a, b)]
        return self.rec.tensor(self.dims, op, ins)

    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)


def _norm_axis(axis, rank):
    return axis + rank if axis < 0 else axis


def make_tf(R):
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.float16, tf.int32, tf.bool = "float32", "float16", "int32", "bool"
    tf.AUTO_REUSE = "AUTO_REUSE"

    class _Scope(object):
        def __init__(self, name):
            self.name = name

    @contextlib.contextmanager
    def variable_scope(name, reuse=None, **_kw):
        if isinstance(name, _Scope):        # tf.variable_scope(tf.get_variable_scope(), ...)
            yield name
            return
        R.scopes.append(name)
        try:
            yield _Scope(R.scope())
        finally:
            R.scopes.pop()

    @contextlib.contextmanager
    def device(_name):
        yield

    def get_variable(name, shape, initializer=None, dtype=None, **_kw):
        full = R.scope() + "/" + name
        R.variables.append({"name": full, "shape": [int(s) for s in shape]})
        return R.tensor(list(shape), "variable", (), name=full)

    def placeholder(dtype=None, shape=None, name=None):
        return R.tensor(list(shape if shape is not None else []), "placeholder", (), dtype=str(dtype), name=name)

    def slice_(x, begin, size):
        dims = [d if s == -1 else s for d, s in zip(x.dims, size)]
        return R.tensor(dims, "slice", [x], begin=list(begin), size=list(size))

    def concat(values=None, axis=None, **kw):
        if values is not None and not isinstance(values, (list, tuple)):     # tf.concat(axis, values) never used; guard anyway
            values, axis = axis, values
        values = kw.get("values", values)
        rank = len(values[0].dims)
        ax = _norm_axis(axis, rank)
        width = [v.dims[ax] for v in values]
        dims = list(values[0].dims)
        dims[ax] = sum(width) if all(isinstance(w, int) for w in width) else None
        return R.tensor(dims, "concat", list(values), axis=ax, widths=width)

    def expand_dims(x, axis):
        ax = axis if axis >= 0 else axis + len(x.dims) + 1
        return R.tensor(x.dims[:ax] + [1] + x.dims[ax:], "expand_dims", [x], axis=ax)

    def _sym(v):          # a dimension argument: an int, or "t<id>" for a tensor-valued one (tf.shape(x)[i])
        return int(v) if isinstance(v, int) else "t%d" % v.id

    def tile(x, multiples):
        mult = [m if isinstance(m, int) else None for m in multiples]
        dims = [(d * m if isinstance(d, int) and isinstance(m, int) else None) for d, m in zip(x.dims, mult)]
        return R.tensor(dims, "tile", [x], multiples=[_sym(m) for m in multiples])

    def reshape(x, shape):
        dims = [s if isinstance(s, int) else None for s in shape]
        return R.tensor(dims, "reshape", [x], to=[_sym(v) for v in shape])

    def squeeze(x, axis=None):
        ax = [_norm_axis(a, len(x.dims)) for a in (axis or [])]
        return R.tensor([d for i, d in enumerate(x.dims) if i not in ax], "squeeze", [x], axis=ax)

    def _reduce(op):
        def f(x, axis=None, keepdims=False, keep_dims=False, name=None):
            ax = [axis] if isinstance(axis, int) else list(axis)
            ax = [_norm_axis(a, len(x.dims)) for a in ax]
            keep = keepdims or keep_dims
            dims = [(1 if i in ax else d) for i, d in enumerate(x.dims) if keep or i not in ax]
            return R.tensor(dims, op, [x], axis=ax, keepdims=bool(keep), name=name)
        return f

    def maximum(x, y):
        return R.tensor(x.dims, "maximum", [x, y if isinstance(y, Tensor) else "const:%r" % (y,)])

    def shape(x):
        return [R.tensor([], "shape_dim", [x], dim=i) for i in range(len(x.dims))]

    def range_(n):
        return R.tensor([None], "range", ["t%d" % n.id if isinstance(n, Tensor) else int(n)])

    def constant(v, dtype=None, **_kw):
        import numpy as np
        return R.tensor(list(np.shape(v)), "constant", (), value=np.asarray(v).tolist())

    def transpose(x, perm):
        return R.tensor([x.dims[p] for p in perm], "transpose", [x], perm=list(perm))

    def cond(pred, true_fn, false_fn):
        # is_training = False (lib/network.py:321 feeds False at test time): the false branch is the inference graph
        out = false_fn()
        R.records.append({"op": "cond(is_training=False)", "scope": R.scope(), "in": [pred.id if isinstance(pred, Tensor) else repr(pred)],
                          "out": out.id if isinstance(out, Tensor) else None, "taken": "false_fn"})
        return out

    def identity(x):
        return x

    tf.variable_scope, tf.device, tf.get_variable, tf.placeholder = variable_scope, device, get_variable, placeholder
    tf.get_variable_scope = lambda: _Scope(R.scope())
    tf.slice, tf.concat, tf.expand_dims, tf.tile, tf.reshape, tf.squeeze = slice_, concat, expand_dims, tile, reshape, squeeze
    tf.reduce_max, tf.reduce_sum, tf.reduce_mean = _reduce("reduce_max"), _reduce("reduce_sum"), _reduce("reduce_mean")
    tf.maximum, tf.shape, tf.range, tf.constant, tf.transpose, tf.cond, tf.identity = maximum, shape, range_, constant, transpose, cond, identity
    tf.multiply = lambda a, b, name=None: a * b
    tf.add_to_collection = lambda *a, **k: None
    tf.truncated_normal_initializer = lambda **k: "truncated_normal"
    tf.constant_initializer = lambda v=0: "constant(%r)" % (v,)

    nn = types.ModuleType("tensorflow.nn")

    def _act(name):
        def f(x, axis=None, name=None):
            return R.tensor(x.dims, name_, [x], axis=None if axis is None else _norm_axis(axis, len(x.dims)))
        name_ = name
        f.__name__ = name
        return f

    nn.relu, nn.sigmoid, nn.tanh, nn.softmax = _act("relu"), _act("sigmoid"), _act("tanh"), _act("softmax")

    def conv2d(x, kernel, strides, padding=None, data_format="NHWC", **_kw):
        assert data_format == "NHWC", data_format
        kh, kw_, cin, cout = kernel.dims
        assert x.dims[-1] == cin, (x.dims, kernel.dims)
        return R.tensor(x.dims[:-1] + [cout], "conv2d", [x, kernel], kernel=[kh, kw_], cin=cin, cout=cout, strides=list(strides),
                        padding=padding, data_format=data_format)

    def conv1d(x, kernel, stride=1, padding=None, data_format="NWC", **_kw):
        assert data_format == "NWC", data_format
        k, cin, cout = kernel.dims
        assert x.dims[-1] == cin, (x.dims, kernel.dims)
        return R.tensor(x.dims[:-1] + [cout], "conv1d", [x, kernel], kernel=[k], cin=cin, cout=cout, stride=stride, padding=padding,
                        data_format=data_format)

    def bias_add(x, b, data_format=None):
        return R.tensor(x.dims, "bias_add", [x, b], data_format=data_format)

    def dropout(x, keep_prob, noise_shape=None):
        return R.tensor(x.dims, "dropout", [x], keep_prob=keep_prob)

    nn.conv2d, nn.conv1d, nn.bias_add, nn.dropout = conv2d, conv1d, bias_add, dropout
    nn.l2_loss = lambda v: v
    tf.nn = nn

    contrib = types.ModuleType("tensorflow.contrib")
    layers = types.ModuleType("tensorflow.contrib.layers")
    layers.xavier_initializer = lambda: "xavier"

    def batch_norm(x, center=True, scale=True, is_training=None, decay=None, updates_collections=None, scope=None, data_format="NHWC", **kw):
        # TensorFlow 1.x knowledge (not read from the reference): the layer owns beta (center), gamma (scale), moving_mean and
        # moving_variance under `scope`; epsilon is the function's default 0.001 unless passed
        c = x.dims[-1] if data_format == "NHWC" else x.dims[1]
        with variable_scope(scope):
            names = (["beta"] if center else []) + (["gamma"] if scale else []) + ["moving_mean", "moving_variance"]
            vs = [get_variable(n, [c]) for n in names]
            return R.tensor(x.dims, "batch_norm", [x] + vs, center=center, scale=scale, epsilon=kw.get("epsilon", 0.001),
                            data_format=data_format, is_training="placeholder" if isinstance(is_training, Tensor) else repr(is_training))

    layers.batch_norm = batch_norm
    contrib.layers = layers
    tf.contrib = contrib
    return tf


def make_native_ops(R):
    """Recorders for the tf.load_op_library wrappers (tf_ops/*/tf_*.py): same names and argument order."""
    samp, grp, itp = types.ModuleType("tf_sampling"), types.ModuleType("tf_grouping"), types.ModuleType("tf_interpolate")

    def farthest_point_sample(npoint, inp):
        return R.tensor([inp.dims[0], npoint], "farthest_point_sample", [inp], npoint=npoint)

    def gather_point(inp, idx):
        return R.tensor([inp.dims[0], idx.dims[1], 3], "gather_point", [inp, idx])

    def query_ball_point(radius, nsample, xyz1, xyz2):
        idx = R.tensor([xyz1.dims[0], xyz2.dims[1], nsample], "query_ball_point", [xyz1, xyz2], radius=radius, nsample=nsample)
        cnt = R.tensor([xyz1.dims[0], xyz2.dims[1]], "query_ball_point.pts_cnt", [idx])
        return idx, cnt

    def group_point(points, idx):
        return R.tensor([points.dims[0], idx.dims[1], idx.dims[2], points.dims[2]], "group_point", [points, idx])

    def knn_point(k, xyz1, xyz2):
        raise AssertionError("knn_point is off the ANCSH graph (knn=False everywhere)")

    def three_nn(xyz1, xyz2):
        d = R.tensor([xyz1.dims[0], xyz1.dims[1], 3], "three_nn.dist", [xyz1, xyz2])
        i = R.tensor([xyz1.dims[0], xyz1.dims[1], 3], "three_nn.idx", [xyz1, xyz2])
        return d, i

    def three_interpolate(points, idx, weight):
        return R.tensor([points.dims[0], idx.dims[1], points.dims[2]], "three_interpolate", [points, idx, weight])

    samp.farthest_point_sample, samp.gather_point = farthest_point_sample, gather_point
    grp.query_ball_point, grp.group_point, grp.knn_point = query_ball_point, group_point, knn_point
    itp.three_nn, itp.three_interpolate = three_nn, three_interpolate
    return {"tf_sampling": samp, "tf_grouping": grp, "tf_interpolate": itp}


FLAGS = {    # main.py:42-52 (argparse store_true defaults are False; only the 'ancsh' branch switches the four flags on), lib/network.py:35-38
    "ancsh": dict(mixed_pred=True, pred_joint=True, pred_joint_ind=True, early_split=True, early_split_nocs=True),
    "npcs": dict(mixed_pred=False, pred_joint=False, pred_joint_ind=False, early_split=False, early_split_nocs=False),
}


def trace(nocs_type, K, N):
    """Run the reference's graph builders under the recorder.  Returns {records, variables, pred (key -> tensor id)}."""
    R = Recorder()
    saved = {k: sys.modules.get(k) for k in ("tensorflow", "tf_sampling", "tf_grouping", "tf_interpolate", "tf_util", "pointnet_util",
                                             "pointnet_plusplus", "pointnet_plusplus.architectures", "pointnet_plusplus.utils",
                                             "pointnet_plusplus.utils.tf_util", "lib", "lib.architecture", "lib.tf_wrapper", "lib.loss")}
    path = list(sys.path)
    try:
        for k in saved:
            sys.modules.pop(k, None)
        sys.modules["tensorflow"] = make_tf(R)
        sys.modules.update(make_native_ops(R))
        # lib/architecture.py imports two modules it does not use on this path (batched_gather; the loss functions): empty stand-ins
        lib = types.ModuleType("lib")
        lib.__path__ = [os.path.join(REF, "lib")]
        sys.modules["lib"] = lib
        tw = types.ModuleType("lib.tf_wrapper")
        tw.batched_gather = None
        sys.modules["lib.tf_wrapper"] = tw
        sys.modules["lib.loss"] = types.ModuleType("lib.loss")
        lib.loss = sys.modules["lib.loss"]
        sys.path.insert(0, REF)
        sys.path.insert(0, os.path.join(REF, "pointnet_plusplus", "utils"))
        import importlib
        arch = importlib.import_module("lib.architecture")
        tf = sys.modules["tensorflow"]
        P = tf.placeholder(dtype=tf.float32, shape=[None, N, 3], name="P")            # lib/network.py:45 ([None, None, 3]; N fixed here so widths resolve)
        is_training = tf.placeholder(dtype=tf.bool, shape=[], name="is_training")      # :43
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            pred = arch.get_per_point_model_new(scope="SPFN", P=P, n_max_parts=K, is_training=is_training, bn_decay=None, **FLAGS[nocs_type])
        return {"nocs_type": nocs_type, "n_max_parts": K, "num_points": N, "flags": FLAGS[nocs_type], "records": R.records,
                "variables": R.variables, "pred": {k: v.id for k, v in pred.items()}}
    finally:
        sys.path[:] = path
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def main():
    out = {"generator": "tests/golden/gen_graph_trace_golden.py",
           "note": "wiring only: produced by the reference's graph-building files under a recording tensorflow stand-in that computes nothing",
           "traces": [trace("ancsh", 3, 1024), trace("npcs", 3, 1024), trace("ancsh", 2, 2048), trace("ancsh", 4, 2048)]}
    with open(os.path.join(HERE, "graph_trace.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    for t in out["traces"]:
        print(t["nocs_type"], t["n_max_parts"], len(t["records"]), "records", len(t["variables"]), "variables")


if __name__ == "__main__":
    main()
