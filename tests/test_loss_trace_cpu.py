"""CPU: tests/golden/loss_trace.json -- the op trace lib/loss.py + lib/network.py (compute_loss, collect_losses) leave under a recording
tensorflow stand-in (tests/golden/gen_loss_trace_golden.py; computes nothing) -- interpreted here in numpy float32 and compared with
oracle/loss_oracle.py on the same inputs.

What this pins is WIRING: which head meets which ground truth under which mask, the flags compute_loss passes (MULTI_HEAD, SELF_SU,
confidence), that the Hungarian matching never reaches a loss, and which multiplier scales which term of total_loss.  The meaning of each
recorded op (below) is TensorFlow 1.x knowledge, not something read from the reference: tf.norm = sqrt(reduce_sum(square)),
tf.one_hot(-1) = zero row, tf.split = equal contiguous parts, reduce_mean = sum / extent in float32."""
import json
import os

import numpy as np
import pytest

from oracle import loss_oracle as LO
from test_loss_cpu import fake_batch

HERE = os.path.dirname(os.path.abspath(__file__))
TRACES = json.load(open(os.path.join(HERE, "golden", "loss_trace.json")))["traces"]
F = np.float32

# ground-truth placeholder -> key of the batch record (lib/network.py:365-382 fill_gt_dict_with_batch_data)
GT_FEED = {"nocs_per_point": "nocs_gt", "cls_per_point": "cls_gt", "mask_array_per_point": "mask_array", "gocs_per_point": "nocs_gt_g",
           "heatmap_per_point": "heatmap_gt", "unitvec_per_point": "unitvec_gt", "orient_per_point": "orient_gt",
           "index_per_point": "joint_cls_gt", "joint_cls_mask": "joint_cls_mask"}


def _const(s):
    v = eval(s[len("const:"):], {"__builtins__": {}})      # 'const:1e-10', 'const:0', 'const:1.0'
    return F(v) if isinstance(v, float) else v


class Interp(object):
    """lazy evaluation of the recorded DAG; `touched` = every record an output depended on"""
    def __init__(self, trace, feed):
        self.rec = {r["out"]: r for r in trace["records"] if r.get("out") is not None}
        self.val, self.touched = dict(feed), set()

    def arg(self, a):
        return _const(a) if isinstance(a, str) else self.get(a)

    def get(self, i):
        if i in self.val:
            return self.val[i]
        r = self.rec[i]
        self.touched.add(i)
        op, x = r["op"], [self.arg(a) for a in r["in"] if not isinstance(a, list)]
        ax = tuple(r["axis"]) if isinstance(r.get("axis"), list) else None
        if op == "shape_dim":
            v = x[0].shape[r["dim"]]
        elif op in ("sub", "add", "mul", "div"):
            v = {"sub": np.subtract, "add": np.add, "mul": np.multiply, "div": np.divide}[op](x[0], x[1])
            v = v.astype(F) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v
        elif op == "split":
            v = np.split(x[0], r["num"], axis=r["axis"])[r["index"]]
        elif op == "norm":
            v = np.sqrt(np.sum(x[0] * x[0], axis=ax, dtype=F))
        elif op == "abs":
            v = np.abs(x[0])
        elif op == "reduce_sum":
            v = np.sum(x[0], axis=ax, dtype=F)
        elif op == "reduce_mean":
            v = np.mean(x[0], axis=ax, dtype=F)        # axis None = 'all'
        elif op == "squeeze":
            v = np.squeeze(x[0], axis=ax)
        elif op == "getitem":
            v = x[0][tuple(slice(None) if k == ":" else k for k in r["key"])]
        elif op == "one_hot":
            depth = r["depth"] if isinstance(r["depth"], int) else int(self.get(int(r["depth"][1:])))
            v = (x[0][..., None] == np.arange(depth)).astype(F)
        elif op == "zeros":
            v = np.zeros(r["shape"], F)
        else:
            raise AssertionError("op %r reaches a loss output but has no meaning here" % op)
        self.val[i] = v
        return v


def run(trace, pred, gt):
    feed = {i: np.asarray(pred[k]) for k, i in trace["pred"].items() if k in pred}
    feed.update({i: np.asarray(gt[GT_FEED[k]]) for k, i in trace["gt"].items() if k in GT_FEED and GT_FEED[k] in gt})
    it = Interp(trace, feed)
    ld = {k: it.get(i) for k, i in trace["loss_dict"].items()}
    tot = {k: it.get(i) for k, i in trace["totals"].items()}
    return ld, tot, it


@pytest.mark.parametrize("ti", range(len(TRACES)))
@pytest.mark.parametrize("seed", [2, 3])        # odd seeds draw cls_gt = -1 (unassigned points: the zero one-hot row)
def test_oracle_equals_the_interpreted_reference_trace(ti, seed):
    t = TRACES[ti]
    K, mixed = t["n_max_parts"], t["flags"]["is_mixed"]
    pred, gt = fake_batch(3, 257, K, seed=seed, mixed=mixed)
    ld, tot, it = run(t, pred, gt)
    want = LO.loss_dict(pred, gt, K, mixed, t["config"]["coord_regress_loss"])
    assert sorted(ld) == sorted(want)
    for k in want:
        assert ld[k].shape == want[k].shape and ld[k].dtype == np.float32, k
        np.testing.assert_allclose(ld[k], want[k], rtol=1e-6, atol=1e-7, err_msg=k)
    c = t["config"]
    mult = dict(miou=c["miou_loss_multiplier"], nocs=c["nocs_loss_multiplier"], gocs=c["gocs_loss_multiplier"], offset=c["offset_loss_multiplier"],
                orient=c["orient_loss_multiplier"], index=c["index_loss_multiplier"], total=c["total_loss_multiplier"])
    assert mult == LO.MULTIPLIERS                               # the yaml the reference's NetworkConfig read at generation time
    wt = LO.collect_losses(want, mixed, t["flags"]["pred_joint"], t["flags"]["pred_joint_ind"], mult)
    assert sorted(tot) == sorted(wt)
    for k in wt:
        assert abs(float(tot[k]) - wt[k]) <= 2e-6 * max(1.0, abs(wt[k])), (k, float(tot[k]), wt[k])
    assert np.float32(t["DIVISION_EPS"]) == LO.DIVISION_EPS


@pytest.mark.parametrize("ti", range(len(TRACES)))
def test_what_never_reaches_a_loss(ti):
    """The Hungarian matching (py_func), the part-presence mask (sequence_mask), the confidence head and joint_params_gt are built by
    compute_loss and consumed by no loss: the oracle's 'no reordering, no confidence' is the reference's wiring, not an assumption."""
    t = TRACES[ti]
    K, mixed = t["n_max_parts"], t["flags"]["is_mixed"]
    pred, gt = fake_batch(2, 64, K, seed=1, mixed=mixed)
    ld, tot, it = run(t, pred, gt)          # neither confi_per_point nor joint_params_gt was fed: evaluation would have raised KeyError
    ops = {t_["op"] for t_ in t["records"]}
    assert {"py_func", "sequence_mask", "stop_gradient"} <= ops
    reached = {it.rec[i]["op"] for i in it.touched}
    assert reached <= {"shape_dim", "sub", "add", "mul", "div", "split", "norm", "abs", "reduce_sum", "reduce_mean", "squeeze", "getitem",
                       "one_hot", "zeros"}, reached
    assert t["matching_indices"] not in it.touched
    # the terms of total_loss: (multiplier, total_*) pairs in the order collect_losses adds them
    byout = {r["out"]: r for r in t["records"] if r.get("out") is not None}
    names = {v: k for k, v in t["totals"].items()}
    terms, cur = [], byout[t["totals"]["total_loss"]]
    assert cur["op"] == "mul" and cur["in"][1] == "const:%r" % t["config"]["total_loss_multiplier"]       # total_loss *= total multiplier
    cur = byout[cur["in"][0]]
    while cur["op"] == "add":
        m = byout[cur["in"][1]]
        assert m["op"] == "mul"
        terms.append((names[m["in"][1]], _const(m["in"][0])))
        cur = byout[cur["in"][0]]
    assert cur["op"] == "zeros"
    c = t["config"]
    want = [("total_nocs_loss", c["nocs_loss_multiplier"]), ("total_miou_loss", c["miou_loss_multiplier"])]
    if mixed:
        want.append(("total_gocs_loss", c["gocs_loss_multiplier"]))
    if t["flags"]["pred_joint"]:
        if mixed:
            want += [("total_heatmap_loss", c["offset_loss_multiplier"]), ("total_unitvec_loss", c["offset_loss_multiplier"])]
        want.append(("total_orient_loss", c["orient_loss_multiplier"]))
        if t["flags"]["pred_joint_ind"]:
            want.append(("total_index_loss", c["index_loss_multiplier"]))
    assert [(n, float(m)) for n, m in terms[::-1]] == [(n, float(F(m))) for n, m in want]


def test_negative_control_a_wrong_mask_is_seen():
    """feed the masks of the parts in the wrong order: the interpreted trace and the oracle (given the right order) must disagree"""
    t = TRACES[0]
    K = t["n_max_parts"]
    pred, gt = fake_batch(2, 128, K, seed=4, mixed=True)
    want = LO.loss_dict(pred, gt, K, True, "L2")
    bad = dict(gt, mask_array=np.ascontiguousarray(gt["mask_array"][:, :, ::-1]))
    ld, _, _ = run(t, pred, bad)
    assert np.abs(ld["nocs_loss"] - want["nocs_loss"]).max() > 1e-3
    np.testing.assert_allclose(ld["miou_loss"], want["miou_loss"], rtol=1e-6, atol=1e-7)


def test_head_widths_agree_with_the_graph_trace():
    """the widths the generator gave the prediction placeholders are the ones the reference's graph builders produce (graph_trace.json)"""
    g = json.load(open(os.path.join(HERE, "golden", "graph_trace.json")))["traces"]
    for t in TRACES:
        gt_ = next(x for x in g if x["nocs_type"] == t["nocs_type"] and x["n_max_parts"] == t["n_max_parts"])
        shape_of = {r["out"]: r["shape"] for r in gt_["records"] if r.get("out") is not None and "shape" in r}
        ph = {r["out"]: r["shape"] for r in t["records"] if r["op"] == "placeholder"}
        for k, i in t["pred"].items():
            if k in gt_["pred"]:
                assert shape_of[gt_["pred"][k]][-1] == ph[i][-1], (t["nocs_type"], k)
