"""Generator of tests/golden/loss_trace.json -- BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

What it pins: the WIRING of the test-time losses (test_loss.txt), taken from the reference's own files executed unmodified:
    lib/loss.py:54-182        compute_nocs_loss / compute_vect_loss / compute_miou_loss
    lib/network.py:340-363    Network.create_gt_dict      (the ground-truth placeholders)
    lib/network.py:421-498    Network.compute_loss        (which loss is fed which head and which mask, with which flags)
    lib/network.py:117-171    Network.collect_losses      (batch means, multipliers, what enters total_loss)
    lib/network_config.py     NetworkConfig getters over cfg/network_config.yml (multipliers, coord_regress_loss)
called the way lib/network.py:70-77 calls them (is_eval=False, is_nn=True), for --nocs_type ancsh and npcs.

What it does NOT pin: arithmetic.  TensorFlow (tensorflow-gpu 1.10.1) and keras are absent from this image and from /root/reference;
the `tensorflow` module these files import here is a RECORDER that computes nothing: a tensor is an (id, static shape) pair and every
op appends one record {op, in (ids or constants, in operand order), out, shape, attributes} to the trace in call order.  What each op
MEANS (tf.norm = sqrt(sum(square)), tf.one_hot(-1) = zero row, reduce_mean over an axis, tf.split into equal parts) is TensorFlow 1.x
knowledge that lives in the interpreter of tests/test_loss_trace_cpu.py, not something read from the reference.

The trace removes the failure mode a self-written oracle cannot see: oracle/loss_oracle.py restating the wrong wiring (a mask on the wrong
head, a confidence where there is none, the Hungarian reordering applied, a multiplier on the wrong term).

    python tests/golden/gen_loss_trace_golden.py          # rewrites tests/golden/loss_trace.json
"""
import contextlib
import io
import json
import os
import sys
import types

REF = os.environ.get("ANCSH_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


class Dim(object):
    """tf.Dimension: compares with ints, and (TF 1.x defines __index__ / __int__) can drive range() and num_or_size_splits"""
    def __init__(self, v):
        self.value = v

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dim) else o)

    def __hash__(self):
        return hash(self.value)

    def __index__(self):
        return int(self.value)

    __int__ = __index__


class Shape(object):
    def __init__(self, dims):
        self.dims = list(dims)

    def __getitem__(self, i):
        return Dim(self.dims[i])

    def as_list(self):
        return list(self.dims)

    def __len__(self):
        return len(self.dims)


def _ref(x):
    """an operand of a record: tensor id, or a python constant spelled out"""
    if isinstance(x, Tensor):
        return x.id
    if isinstance(x, Dim):
        return "const:%r" % (x.value,)
    if isinstance(x, (list, tuple)):
        return [_ref(v) for v in x]
    return "const:%r" % (x,)


def _attr(v):
    if isinstance(v, Tensor):
        return "t%d" % v.id
    if isinstance(v, Dim):
        return v.value
    if isinstance(v, (list, tuple)):
        return [_attr(x) for x in v]
    if callable(v):
        return "fn:%s" % getattr(v, "__name__", "?")
    return v if isinstance(v, (int, float, str, bool)) or v is None else repr(v)


class Recorder(object):
    def __init__(self):
        self.records, self.n = [], 0

    def tensor(self, shape, op, inputs=(), **attrs):
        self.n += 1
        t = Tensor(self, self.n, shape)
        rec = {"op": op, "in": [_ref(x) for x in inputs], "out": t.id, "shape": [d if isinstance(d, int) else None for d in shape]}
        rec.update({k: _attr(v) for k, v in attrs.items() if v is not None})
        self.records.append(rec)
        return t


def _bshape(a, b):
    """static shape of an elementwise op with numpy/TF broadcasting (None = unknown extent)"""
    da = a.dims if isinstance(a, Tensor) else []
    db = b.dims if isinstance(b, Tensor) else []
    out = []
    for i in range(max(len(da), len(db))):
        x = da[len(da) - 1 - i] if i < len(da) else 1
        y = db[len(db) - 1 - i] if i < len(db) else 1
        out.append(y if x == 1 else x if y == 1 or x == y else (x if y is None else y))
    return out[::-1]


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
        return self.rec.tensor(_bshape(a, b), op, [a, b])

    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return self.rec.tensor(self.dims, "neg", [self])
    def __gt__(self, o): return self._bin("greater", o)

    def __getitem__(self, key):
        key = key if isinstance(key, tuple) else (key,)
        dims, spec = [], []
        for i, k in enumerate(key):
            if isinstance(k, slice):
                assert k == slice(None), "only full slices appear on this path"
                dims.append(self.dims[i])
                spec.append(":")
            else:
                spec.append(int(k))
        dims += self.dims[len(key):]
        return self.rec.tensor(dims, "getitem", [self], key=spec)


def _norm_axis(axis, rank):
    return axis + rank if axis < 0 else axis


def make_tf(R):
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.int32, tf.bool = "float32", "int32", "bool"

    def placeholder(dtype=None, shape=None, name=None):
        return R.tensor(list(shape if shape is not None else []), "placeholder", (), dtype=str(dtype))

    def shape(x):
        return [R.tensor([], "shape_dim", [x], dim=i) for i in range(len(x.dims))]

    def _reduce(op):
        def f(x, axis=None, keepdims=False, keep_dims=False, name=None):
            if axis is None:
                return R.tensor([], op, [x], axis="all")
            ax = [_norm_axis(a, len(x.dims)) for a in ([axis] if isinstance(axis, int) else list(axis))]
            keep = keepdims or keep_dims
            dims = [(1 if i in ax else d) for i, d in enumerate(x.dims) if keep or i not in ax]
            return R.tensor(dims, op, [x], axis=ax, keepdims=bool(keep))
        return f

    def split(value, num_or_size_splits, axis=0):
        n = int(num_or_size_splits)
        ax = _norm_axis(axis, len(value.dims))
        w = value.dims[ax]
        assert isinstance(w, int) and w % n == 0, (value.dims, n)
        dims = list(value.dims)
        dims[ax] = w // n
        return [R.tensor(dims, "split", [value], num=n, axis=ax, index=i) for i in range(n)]

    def norm(x, ord="euclidean", axis=None, **_kw):
        assert ord == "euclidean" and isinstance(axis, int)
        ax = _norm_axis(axis, len(x.dims))
        return R.tensor([d for i, d in enumerate(x.dims) if i != ax], "norm", [x], axis=[ax], ord=ord)

    def squeeze(x, axis=None):
        ax = [_norm_axis(a, len(x.dims)) for a in ([axis] if isinstance(axis, int) else list(axis or []))]
        return R.tensor([d for i, d in enumerate(x.dims) if i not in ax], "squeeze", [x], axis=ax)

    def one_hot(indices, depth, dtype=None, **_kw):
        return R.tensor(indices.dims + [depth if isinstance(depth, int) else None], "one_hot", [indices], depth=depth, dtype=str(dtype))

    def zeros(shape=None, dtype=None):
        return R.tensor(list(shape), "zeros", (), dtype=str(dtype))

    def _unary(op):
        return lambda x, name=None: R.tensor(x.dims, op, [x])

    def generic(name):
        # every other tf.* the files touch on the way (sequence_mask, stop_gradient, py_func, reduce_max ...): recorded with its
        # operands and keyword arguments, shape unknown -- the test asserts which of them reach a loss output (none may)
        def f(*args, **kw):
            ins = [a for a in args if isinstance(a, (Tensor, list, tuple))]
            rest = [a for a in args if not isinstance(a, (Tensor, list, tuple))]
            return R.tensor([], name, ins, args=rest, **kw)
        return f

    tf.placeholder, tf.shape, tf.split, tf.norm, tf.squeeze, tf.one_hot, tf.zeros = placeholder, shape, split, norm, squeeze, one_hot, zeros
    tf.reduce_sum, tf.reduce_mean, tf.reduce_max = _reduce("reduce_sum"), _reduce("reduce_mean"), _reduce("reduce_max")
    tf.abs, tf.log, tf.to_float, tf.zeros_like = _unary("abs"), _unary("log"), _unary("to_float"), _unary("zeros_like")
    summary = types.ModuleType("tensorflow.summary")
    summary.scalar = lambda name, t: R.records.append({"op": "summary.scalar", "in": [_ref(t)], "out": None, "name": name})
    tf.summary = summary
    tf.__dict__["__getattr__"] = generic          # PEP 562: any tf.<name> not defined above
    return tf


FLAGS = {    # main.py:42-52 + lib/network.py:31-38: what the two --nocs_type values switch on
    "ancsh": dict(is_mixed=True, pred_joint=True, pred_joint_ind=True),
    "npcs": dict(is_mixed=False, pred_joint=False, pred_joint_ind=False),
}


def pred_placeholders(tf, K, is_mixed):
    """The heads as lib/architecture.py:86-161,195-208 shapes them (tests/golden/graph_trace.json holds the same widths; the test
    cross-checks): B and N unknown, the channel width static -- compute_vect_loss branches on vect.shape[2] == 1."""
    p = {"W": tf.placeholder(tf.float32, [None, None, K]), "nocs_per_point": tf.placeholder(tf.float32, [None, None, 3 * K]),
         "confi_per_point": tf.placeholder(tf.float32, [None, None, 1]), "heatmap_per_point": tf.placeholder(tf.float32, [None, None, 1]),
         "unitvec_per_point": tf.placeholder(tf.float32, [None, None, 3]), "joint_axis_per_point": tf.placeholder(tf.float32, [None, None, 3]),
         "index_per_point": tf.placeholder(tf.float32, [None, None, 3])}       # joint_est_model: three joint classes whatever K is
    if is_mixed:
        p["gocs_per_point"] = tf.placeholder(tf.float32, [None, None, 3 * K])
    return p


def trace(nocs_type, K):
    R = Recorder()
    names = ("tensorflow", "keras", "keras.backend", "lib", "lib.tf_wrapper", "constants", "loss", "architecture", "prediction_io",
             "network", "network_config", "_init_paths", "global_info")
    saved = {k: sys.modules.get(k) for k in names}
    path = list(sys.path)
    try:
        for k in saved:
            sys.modules.pop(k, None)
        tf = make_tf(R)
        sys.modules["tensorflow"] = tf
        keras = types.ModuleType("keras")                  # imported by lib/loss.py:9-10, unused on this path
        keras.backend = types.ModuleType("keras.backend")
        sys.modules["keras"], sys.modules["keras.backend"] = keras, keras.backend
        lib = types.ModuleType("lib")
        lib.__path__ = [os.path.join(REF, "lib")]
        sys.modules["lib"] = lib
        tw = types.ModuleType("lib.tf_wrapper")            # lib/loss.py:5: batched_gather only serves matching_indices (never passed)

        def batched_gather(*a, **k):
            raise AssertionError("batched_gather is off the test-time path (compute_miou_loss is called without matching_indices)")

        tw.batched_gather = batched_gather
        sys.modules["lib.tf_wrapper"] = tw
        # lib/network.py:6,9 import the graph builder and the .h5 writer: neither is touched by the three methods traced here
        sys.modules["architecture"] = types.ModuleType("architecture")
        sys.modules["prediction_io"] = types.ModuleType("prediction_io")
        sys.modules["_init_paths"] = types.ModuleType("_init_paths")
        gi = types.ModuleType("global_info")
        gi.global_info = lambda: types.SimpleNamespace(base_path="")
        sys.modules["global_info"] = gi
        sys.path.insert(0, os.path.join(REF, "lib"))
        sys.path.insert(0, REF)
        import importlib
        with contextlib.redirect_stdout(io.StringIO()):
            network = importlib.import_module("network")
            ncfg = importlib.import_module("network_config")
        import yaml
        config = object.__new__(ncfg.NetworkConfig)          # its __init__ only parses the command line + this file
        config.conf = yaml.safe_load(open(os.path.join(REF, "cfg", "network_config.yml")))
        f = FLAGS[nocs_type]
        me = types.SimpleNamespace(config=config, **f)
        Net = network.Network
        pred = pred_placeholders(tf, K, f["is_mixed"])
        gt = Net.create_gt_dict(me, K)
        with contextlib.redirect_stdout(io.StringIO()):
            res = Net.compute_loss(me, pred, gt, config, is_eval=False, is_nn=True)
            Net.collect_losses(me, res["loss_dict"])
        totals = {k: getattr(me, k).id for k in ("total_loss", "total_miou_loss", "total_nocs_loss", "total_heatmap_loss", "total_unitvec_loss",
                                                  "total_orient_loss", "total_index_loss") if hasattr(me, k)}
        if f["is_mixed"]:
            totals["total_gocs_loss"] = me.total_gocs_loss.id
        constants = importlib.import_module("constants")
        return {"nocs_type": nocs_type, "n_max_parts": K, "flags": f, "records": R.records,
                "pred": {k: v.id for k, v in pred.items()}, "gt": {k: v.id for k, v in gt.items()},
                "loss_dict": {k: v.id for k, v in res["loss_dict"].items()}, "matching_indices": res["matching_indices"].id,
                "totals": totals, "DIVISION_EPS": float(constants.DIVISION_EPS),
                "config": {k: config.conf[k] for k in ("miou_loss_multiplier", "nocs_loss_multiplier", "gocs_loss_multiplier",
                                                       "offset_loss_multiplier", "orient_loss_multiplier", "index_loss_multiplier",
                                                       "total_loss_multiplier", "coord_regress_loss")}}
    finally:
        sys.path[:] = path
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def main():
    out = {"generator": "tests/golden/gen_loss_trace_golden.py",
           "note": "wiring only: produced by lib/loss.py + lib/network.py (compute_loss, collect_losses) under a recording tensorflow stand-in that computes nothing",
           "traces": [trace("ancsh", 3), trace("npcs", 3), trace("ancsh", 2), trace("ancsh", 4)]}
    with open(os.path.join(HERE, "loss_trace.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    for t in out["traces"]:
        print(t["nocs_type"], t["n_max_parts"], len(t["records"]), "records", sorted(t["loss_dict"]))


if __name__ == "__main__":
    main()
