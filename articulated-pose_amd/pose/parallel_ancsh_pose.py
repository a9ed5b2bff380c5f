"""Counterpart of evaluation/parallel_ancsh_pose.py on the MI355X.

Reference: per cloud, K x ransac(single_*, niter=10000) + (K-1) x ransac(joint_*, niter=200), each
hypothesis a numpy Kabsch fit (+ a scipy LM for joints), ~10 s/cloud/core.  Here a whole batch of clouds
is solved by five kernel launches (partition, joint-direction medians, stage A score + finish, stage B
hypotheses + finish); hypotheses are the parallel axis.

Randomness: the reference samples inside the estimators from numpy's unseeded global RNG.  Here the
3-point samples are an explicit input (`draws_*`, int32) so a run can replay numpy's stream exactly
(`draws_from_seed`), or -- draws None -- come from the on-device counter-based generator (`seed`).
"""
import os
import pickle

import numpy as np
import torch

from .. import _lib
from .d3_utils import rot_diff_degree


def _i32(t, dev):
    if torch.is_tensor(t):
        return t.to(dev, torch.int32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(t, np.int32)).to(dev)


def _f32(t, dev):
    if torch.is_tensor(t):
        return t.to(dev, torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(t, np.float32)).to(dev)


TIE_WINDOW = 32 * 2.0 ** -27      # default half-width of the tie window on the residual norm: 32 ulp of the float32 threshold 0.1 (2.4e-7)


def ransac_single_batch(off, src, tgt, inlier_th, niter, draws=None, seed=0, max_n=None, scalar_points=True, record=None, K=0,
                        tie_window=None):
    """Batched ransac(dataset, single_transformation_estimator, single_transformation_verifier, th, niter).
    off (nprob+1) int32 row offsets into src/tgt (rows,3) float32 device tensors.
    scalar_points: score the hypotheses with the parts' points in scalar registers (ancsh_ransac_single_ex: a padded quad copy of
    the parts in scratch memory, no LDS) instead of the LDS-staged kernel of ancsh_ransac_single; identical results.
    record: optional (B, K, 26) float64 pose record -- every fit is ALSO written into columns 0..12 of its row by the finish kernel
    (ancsh_ransac_single_rec); tie_window: not None -> `tie` (nprob, 2) int32 [borderline points of the winner, degenerate
    contenders] (see include/ancsh_hip.h).
    -> dict(model (nprob,13) f64 [R(9) s t(3)], inliers (rows) uint8, best (nprob,2) int32 [iter, score])."""
    dev = src.device
    nprob = off.numel() - 1
    rows = src.shape[0]
    max_n = int(max_n or rows)
    model = torch.empty((nprob, 13), dtype=torch.float64, device=dev)
    inl = torch.empty((rows,), dtype=torch.uint8, device=dev)      # every row inside off[0]..off[-1] is written by the finish kernel
    best = torch.empty((nprob, 2), dtype=torch.int32, device=dev)
    scores = torch.empty((nprob * niter,), dtype=torch.int32, device=dev)
    d = None if draws is None else _i32(draws, dev)
    if d is not None and d.numel() != nprob * niter * 3:
        raise ValueError("draws must have shape (nprob, niter, 3)")
    quads, tie = None, None
    if scalar_points:
        quads = torch.empty((_lib.lib().ancsh_ransac_single_quads_floats(rows, nprob),), dtype=torch.float32, device=dev)
    if record is not None or tie_window is not None:
        tie = torch.empty((nprob, 2), dtype=torch.int32, device=dev) if tie_window is not None else None
        _lib.call("ancsh_ransac_single_rec", nprob, _lib.ptr(off), _lib.ptr(src), _lib.ptr(tgt), float(inlier_th), int(niter),
                  _lib.ptr(d), int(seed), max_n, _lib.ptr(model), _lib.ptr(inl), _lib.ptr(best), _lib.ptr(scores), _lib.ptr(quads), rows,
                  _lib.ptr(record), int(K), _lib.ptr(tie), float(tie_window or 0.0))
    elif scalar_points:
        _lib.call("ancsh_ransac_single_ex", nprob, _lib.ptr(off), _lib.ptr(src), _lib.ptr(tgt), float(inlier_th), int(niter),
                  _lib.ptr(d), int(seed), max_n, _lib.ptr(model), _lib.ptr(inl), _lib.ptr(best), _lib.ptr(scores), _lib.ptr(quads), rows)
    else:
        _lib.call("ancsh_ransac_single", nprob, _lib.ptr(off), _lib.ptr(src), _lib.ptr(tgt), float(inlier_th), int(niter),
                  _lib.ptr(d), int(seed), max_n, _lib.ptr(model), _lib.ptr(inl), _lib.ptr(best), _lib.ptr(scores))
    return dict(model=model, inliers=inl, best=best, scores=scores.view(nprob, niter), tie=tie, _keep=(d, scores, quads))


LM_SCHEDULES = {"auto": 0, "throughput": 1, "latency": 2}     # ANCSH_LM_* of include/ancsh_hip.h


def ransac_joint_batch(rng0, rng1, src, tgt, joint_dir, inlier_th, niter, draws=None, seed=0, max_n=None, want_lm_stat=False,
                       lm_schedule="auto", record=None, K=0, tie_window=None):
    """Batched ransac(dataset, joint_transformation_estimator, joint_transformation_verifier, th, niter).
    rng0/rng1 (nprob,2) int32 [start,end) rows of part 0 / part j; joint_dir (nprob,3) float32.
    -> dict(model (nprob,26) f64 [R0 s0 t0 R1 s1 t1], inliers (nprob,2,max_n) uint8, best (nprob), score (nprob))."""
    dev = src.device
    nprob = rng0.shape[0]
    max_n = int(max_n or src.shape[0])
    model = torch.empty((nprob, 26), dtype=torch.float64, device=dev)
    inl = torch.empty((nprob, 2, max_n), dtype=torch.uint8, device=dev)   # fully written (flags + zero tail) by the finish kernel
    best = torch.empty((nprob,), dtype=torch.int32, device=dev)
    score = torch.empty((nprob,), dtype=torch.float64, device=dev)
    sc = torch.empty((nprob * niter,), dtype=torch.float64, device=dev)
    mo = torch.empty((nprob * niter, 26), dtype=torch.float64, device=dev)
    stat = torch.empty((nprob, niter, 2), dtype=torch.int32, device=dev) if want_lm_stat else None
    d = None if draws is None else _i32(draws, dev)
    if d is not None and d.numel() != nprob * niter * 6:
        raise ValueError("draws must have shape (nprob, niter, 6)")
    tie = None
    if record is not None or tie_window is not None:      # the finish kernel also fills the record's nonlinear columns / the tie counts
        tie = torch.empty((nprob, 2), dtype=torch.int32, device=dev) if tie_window is not None else None
        _lib.call("ancsh_ransac_joint_rec", nprob, _lib.ptr(rng0), _lib.ptr(rng1), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(joint_dir),
                  float(inlier_th), int(niter), _lib.ptr(d), int(seed), max_n, _lib.ptr(model), _lib.ptr(inl), _lib.ptr(best),
                  _lib.ptr(score), _lib.ptr(sc), _lib.ptr(mo), _lib.ptr(stat), LM_SCHEDULES[lm_schedule], _lib.ptr(record), int(K),
                  _lib.ptr(tie), float(tie_window or 0.0))
    else:
        _lib.call("ancsh_ransac_joint_ex", nprob, _lib.ptr(rng0), _lib.ptr(rng1), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(joint_dir),
                  float(inlier_th), int(niter), _lib.ptr(d), int(seed), max_n, _lib.ptr(model), _lib.ptr(inl), _lib.ptr(best),
                  _lib.ptr(score), _lib.ptr(sc), _lib.ptr(mo), _lib.ptr(stat), LM_SCHEDULES[lm_schedule])
    return dict(model=model, inliers=inl, best=best, score=score, lm_stat=stat, hyp_models=mo, hyp_scores=sc, tie=tie, _keep=(d,))


def draws_from_seed(seed, counts, niter_a, niter_b):
    """Replay np.random.seed(seed) + the reference's randint call order for ONE cloud
    (stage A parts 0..K-1, then joints 1..K-1: evaluation/parallel_ancsh_pose.py:38,110-111).
    counts: points per predicted part.  -> (draws_a (K,niter_a,3), draws_b (K-1,niter_b,6)) int32."""
    rs = np.random.RandomState(seed)
    K = len(counts)
    da = np.zeros((K, niter_a, 3), np.int32)
    for j in range(K):
        for i in range(niter_a):
            da[j, i] = rs.randint(counts[j], size=3)
    db = np.zeros((max(K - 1, 0), niter_b, 6), np.int32)
    for j in range(1, K):
        for i in range(niter_b):
            db[j - 1, i, :3] = rs.randint(counts[0], size=3)
            db[j - 1, i, 3:] = rs.randint(counts[j], size=3)
    return da, db


class PoseSolver(object):
    """solve(...) = the per-cloud body of solver_ransac_nonlinear (:238-341) for a batch of clouds.

    Inputs per cloud (device tensors or arrays): P (B,N,3); nocs_pred (B,N,3K) and mask_pred (B,N,K) (from
    the part-NOCS/baseline network when USE_BASELINE, :232-237); joint_axis_per_point (B,N,3) and
    joint_cls (B,N) int (from the ANCSH record, :295).  Returns device tensors:
        record    (B,K,26) float64  [baseline | nonlinear]: the per-part pose record (:330-353), written by the two finish kernels
        baseline  (B,K,13) float64  [R(9) row-major, scale, translation(3)]   -- stage A   (= record[:, :, :13], a view)
        nonlinear (B,K,13) float64                                             -- stage B (part 0 from joint 1; record[:, :, 13:])
        counts (B,K) int32 points per predicted part; best_a (B,K,2); best_b (B,K-1)
        tie_a (B,K,2), tie_b (B,K-1,2) int32: how implementation-sensitive each fit is -- [points within `tie_window` of the inlier
        threshold under the winning hypothesis, DEGENERATE contenders = hypotheses within one inlier of the winning score whose 3-point
        sample repeats an index (their rotation is implementation-defined in the reference itself); stage A counts only those that would
        change the consensus set and makes the count NEGATIVE when the winner's own sample is degenerate -- the sharp per-fit warning]
        (include/ancsh_hip.h, ancsh_ransac_single_rec; what the counts did and did not predict: profiles/r06_pose_tie_rate_K3.txt)
    A part with no predicted points gives NaN rows (the reference raises inside randint)."""

    def __init__(self, num_parts, inlier_th=0.1, niter_a=10000, niter_b=200, device="cuda:0", want_lm_stat=False,
                 max_part_points=None, lm_schedule="auto", tie_window=TIE_WINDOW):
        self.K, self.th, self.niter_a, self.niter_b = num_parts, inlier_th, niter_a, niter_b
        self.tie_window = tie_window           # None: no tie counts
        self.device = torch.device(device)
        # Upper bound on the points of ONE predicted part (sizes the LDS-resident refits: <= 6144 for stage A, <= 3072 for the
        # joint fit).  Default: the whole cloud N, which needs no host synchronisation (graph capture) and covers N <= 3072;
        # for larger clouds whose parts are known to be smaller, pass the bound here.
        self.max_part_points = max_part_points
        if lm_schedule not in LM_SCHEDULES:
            raise ValueError("lm_schedule must be one of %s" % sorted(LM_SCHEDULES))
        self.lm_schedule = lm_schedule         # "latency": eight lanes per LM fit (a lone batch finishes sooner); "throughput": one
        self.want_lm_stat = want_lm_stat       # also return per-hypothesis (status, nfev) of the stage-B LM fits

    def solve(self, P, nocs_pred, mask_pred, joint_axis_per_point, joint_cls, draws_a=None, draws_b=None, seed=0):
        """Both stages of a batch.  The joint fit (stage B) only needs the partition, not the per-part fits, and it is the
        latency-bound half (64 waves for 1.6 ms: MINPACK's longest trajectory), so it is ISSUED FIRST: its LM kernel then runs
        under the full-chip scoring kernel of other batches in flight, and a batch ends with 0.3 ms of stage A instead of idling
        the chip through the LM tail (what a short run's drain is made of).  Same results as A-then-B: the stages share nothing
        but the partition."""
        out = self._partition(P, nocs_pred, mask_pred)
        self.solve_stage_b(out, joint_axis_per_point, joint_cls, draws_b, seed)
        return self._poison(self._stage_a_fits(out, draws_a, seed))

    def solve_stage_a(self, P, nocs_pred, mask_pred, draws_a=None, seed=0):
        """Part labels + per-part RANSAC / Kabsch (stage A, :238-272): needs only the part-NOCS network's outputs."""
        out = self._partition(P, nocs_pred, mask_pred)
        if self.K > 1:
            out["record"][:, :, 13:] = float("nan")      # stage B never runs on this path: "not fitted", never stale memory
        return self._poison(self._stage_a_fits(out, draws_a, seed))

    def _poison(self, out):
        """A cloud with a NaN / +-Inf anywhere in the fit's inputs gets an all-NaN record (include/ancsh_hip.h,
        ancsh_pose_poison_records): the reference raises LinAlgError on such a cloud; here the batch goes on and the cloud says so."""
        B, N = out["_shape"]
        P, nocs, W = out["_inputs"]
        axis = out.get("_axis")                   # stage B's joint-axis field (K > 1 and stage B ran)
        _lib.call("ancsh_pose_poison_records", B, N, self.K, _lib.ptr(P), _lib.ptr(nocs), _lib.ptr(W), _lib.ptr(axis), _lib.ptr(out["record"]))
        return out

    def _partition(self, P, nocs_pred, mask_pred):
        dev, K = self.device, self.K
        P, nocs, W = _f32(P, dev), _f32(nocs_pred, dev), _f32(mask_pred, dev)
        B, N, _ = P.shape
        if nocs.shape != (B, N, 3 * K) or W.shape != (B, N, K):
            raise ValueError("nocs_pred must be (B,N,3K) and mask_pred (B,N,K)")
        max_n = min(N, self.max_part_points or N)
        labels = torch.empty((B, N), dtype=torch.int32, device=dev)
        pidx = torch.empty((B, N), dtype=torch.int32, device=dev)
        off = torch.empty((B * K + 1,), dtype=torch.int32, device=dev)
        src = torch.empty((B * N, 3), dtype=torch.float32, device=dev)
        tgt = torch.empty((B * N, 3), dtype=torch.float32, device=dev)
        counts = torch.empty((B, K), dtype=torch.int32, device=dev)
        rng0 = torch.empty((B * max(K - 1, 1), 2), dtype=torch.int32, device=dev) if K > 1 else None
        rng1 = torch.empty((B * max(K - 1, 1), 2), dtype=torch.int32, device=dev) if K > 1 else None
        _lib.call("ancsh_pose_partition", B, N, K, _lib.ptr(W), _lib.ptr(P), _lib.ptr(nocs), _lib.ptr(labels), _lib.ptr(pidx),
                  _lib.ptr(off), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(counts), _lib.ptr(rng0), _lib.ptr(rng1))
        # stage A's finish kernel writes columns 0..12 of every row, stage B's 13..25 (K == 1: stage A writes both): solve() runs both, so
        # nothing is filled here (no aten launch in the captured step); solve_stage_a() marks the half it leaves unwritten
        record = torch.empty((B, K, 26), dtype=torch.float64, device=dev)
        return dict(labels=labels, part_index=pidx, off=off, counts=counts, record=record, _src=src, _tgt=tgt, _max_n=max_n, _shape=(B, N),
                    _rng=(rng0, rng1), _inputs=(P, nocs, W))

    def _stage_a_fits(self, out, draws_a=None, seed=0):
        dev, K = self.device, self.K
        B, N = out["_shape"]
        a = ransac_single_batch(out["off"], out["_src"], out["_tgt"], self.th, self.niter_a,
                                None if draws_a is None else _i32(draws_a, dev).reshape(B * K, self.niter_a, 3), seed, out["_max_n"],
                                record=out["record"], K=K, tie_window=self.tie_window)
        out.update(baseline=out["record"][:, :, :13], best_a=a["best"].view(B, K, 2), inliers_a=a["inliers"].view(B, N))
        if a["tie"] is not None:
            out["tie_a"] = a["tie"].view(B, K, 2)
        if K == 1:
            out["nonlinear"] = out["record"][:, :, 13:]           # a one-part object: the finish kernel wrote the baseline there too
        return out

    def solve_stage_b(self, out, joint_axis_per_point, joint_cls, draws_b=None, seed=0):
        """Articulated joint fit (stage B, :274-341) on top of a solve_stage_a result."""
        dev, K = self.device, self.K
        B, N = out["_shape"]
        src, tgt, max_n = out["_src"], out["_tgt"], out["_max_n"]
        rng0, rng1 = out["_rng"]                                   # written by the partition kernel
        if K > 1:
            axis, jcls = _f32(joint_axis_per_point, dev), _i32(joint_cls, dev)
            out["_axis"] = axis
            jdir = torch.empty((B, K - 1, 3), dtype=torch.float32, device=dev)
            _lib.call("ancsh_pose_joint_direction", B, N, K, _lib.ptr(axis), _lib.ptr(jcls), _lib.ptr(jdir))
            b = ransac_joint_batch(rng0, rng1, src, tgt, jdir.view(-1, 3), self.th, self.niter_b,
                                   None if draws_b is None else _i32(draws_b, dev).reshape(B * (K - 1), self.niter_b, 6),
                                   seed + 1, max_n, want_lm_stat=self.want_lm_stat, lm_schedule=self.lm_schedule,
                                   record=out["record"], K=K, tie_window=self.tie_window)
            if self.want_lm_stat:
                out["lm_stat"] = b["lm_stat"].view(B, K - 1, self.niter_b, 2)
            out["nonlinear"] = out["record"][:, :, 13:]
            if b["tie"] is not None:
                out["tie_b"] = b["tie"].view(B, K - 1, 2)
            out["best_b"] = b["best"].view(B, K - 1)
            out["score_b"] = b["score"].view(B, K - 1)          # the winning hypothesis's verifier score (:186-194)
            out["joint_direction"] = jdir
            out["inliers_b"] = b["inliers"].view(B, K - 1, 2, max_n)
        else:
            out["nonlinear"] = out["record"][:, :, 13:]          # K = 1: filled by _stage_a_fits' finish kernel
        return out


def _model_to_rst(m):
    return m[:9].reshape(3, 3), m[9], m[10:13]


def records_from_solution(sol, rts_list=None):
    """Per-cloud dicts with the reference's pickle schema (:346-352): scale / rotation / translation with
    'baseline' and 'nonlinear' lists of length K (+ 'gt' and the error entries when GT is supplied as
    rts_list[b] = {'scale': {'gt': [...]}, 'rt': {'gt': [4x4...]}} from compute_gt_pose.py)."""
    base = sol["baseline"].cpu().numpy()
    nonl = sol["nonlinear"].cpu().numpy()
    B, K, _ = base.shape
    recs = []
    for b in range(B):
        scale_dict = {'gt': [], 'baseline': [], 'nonlinear': []}
        r_dict = {'gt': [], 'baseline': [], 'nonlinear': []}
        t_dict = {'gt': [], 'baseline': [], 'nonlinear': []}
        xyz_err = {'baseline': [], 'nonlinear': []}
        rpy_err = {'baseline': [], 'nonlinear': []}
        scale_err = {'baseline': [], 'nonlinear': []}
        for j in range(K):
            for kind, arr in (('baseline', base), ('nonlinear', nonl)):
                R, s, t = _model_to_rst(arr[b, j])
                scale_dict[kind].append(s)
                r_dict[kind].append(R)
                t_dict[kind].append(t)
                if rts_list is not None:
                    rt_gt, s_gt = rts_list[b]['rt']['gt'][j], rts_list[b]['scale']['gt'][j]
                    rpy_err[kind].append(rot_diff_degree(R, rt_gt[:3, :3]))
                    xyz_err[kind].append(np.linalg.norm(t - rt_gt[:3, 3]))
                    scale_err[kind].append(np.linalg.norm(s - s_gt[0]))
            if rts_list is not None:
                scale_dict['gt'].append(rts_list[b]['scale']['gt'][j][0])
                r_dict['gt'].append(rts_list[b]['rt']['gt'][j][:3, :3])
                t_dict['gt'].append(rts_list[b]['rt']['gt'][j][:3, 3])
        recs.append({'scale': scale_dict, 'rotation': r_dict, 'translation': t_dict, 'xyz_err': xyz_err,
                     'rpy_err': rpy_err, 'scale_err': scale_err})
    return recs


def solver_ransac_nonlinear(s_ind, e_ind, test_exp, baseline_exp, choose_threshold, num_parts, test_group, problem_ins,
                            rts_all, file_name, base_path=None, batch_size=32, seed=0, device="cuda:0"):
    """Same positional signature as the reference entry point (:196): solves test_group[s_ind:e_ind] and
    pickles {basename: record}.  Records are read with prediction_io.load_record from
    <base_path>/results/test_pred/<exp>/<basename>.{h5,npz} (USE_BASELINE: NOCS + mask from baseline_exp)."""
    from .. import prediction_io
    base_path = base_path or os.environ.get("ANCSH_BASE_PATH", ".")
    solver = PoseSolver(num_parts, choose_threshold, device=device)
    names = [test_group[i].split('.')[0] for i in range(s_ind, e_ind) if test_group[i].split('_')[0] not in problem_ins]
    all_rts = {}
    for c0 in range(0, len(names), batch_size):
        chunk = names[c0:c0 + batch_size]
        f = [prediction_io.load_record(os.path.join(base_path, 'results/test_pred', str(test_exp)), n) for n in chunk]
        fb = [prediction_io.load_record(os.path.join(base_path, 'results/test_pred', str(baseline_exp)), n) for n in chunk]
        sol = solver.solve(np.stack([r['P'][:, :3] for r in f]), np.stack([r['nocs_per_point'] for r in fb]),
                           np.stack([r['instance_per_point'] for r in fb]), np.stack([r['joint_axis_per_point'] for r in f]),
                           np.stack([r['joint_cls_gt'] for r in f]), seed=seed + c0)
        gts = [rts_all[n] for n in chunk] if rts_all is not None else None
        for n, rec in zip(chunk, records_from_solution(sol, gts)):
            all_rts[n] = rec
    with open(file_name, 'wb') as fh:
        pickle.dump(all_rts, fh)
    return all_rts
