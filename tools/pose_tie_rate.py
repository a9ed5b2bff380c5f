#!/usr/bin/env python
"""How often do the HIP pose fit and the reference arithmetic crown DIFFERENT RANSAC hypotheses, and what does it do to R, s, t?

    python tools/pose_tie_rate.py [--clouds 700] [--parts 3] [--npoints 1024] [--na 2000] [--nb 64] [--workers W] [--out FILE]

Every cloud (articulated_pose_amd.synthetic, the benchmark's distribution) is solved by the HIP path (PoseSolver, replayed draws)
and by oracle/pose_oracle.py (= the reference's numpy / scipy calls, evaluation/parallel_ancsh_pose.py:20-54,106-194) on the SAME
draws; oracle/pose_compare.py lines the fits up.  Report: fits, promotions (different winning iteration), their score difference in
inliers and the largest |dR|, |ds|, |dt| of the final refits -- for promoted and for agreeing winners.  The oracle side runs on
`--workers` single-threaded CPU processes."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=700)
    ap.add_argument("--parts", type=int, default=3)
    ap.add_argument("--npoints", type=int, default=1024)
    ap.add_argument("--na", type=int, default=2000)
    ap.add_argument("--nb", type=int, default=64)
    ap.add_argument("--first", type=int, default=5000, help="id of the first synthetic cloud")
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--tie-window-ulps", type=float, default=32.0, help="half-width of the tie window in ulps of float32(0.1)")
    a = ap.parse_args()
    import torch
    import articulated_pose_amd  # noqa: F401
    from articulated_pose_amd.pose import PoseSolver
    from oracle import cpu_layout, pose_compare as PC
    K, N = a.parts, a.npoints
    workers = a.workers or max(1, cpu_layout.usable_cpus() - 2)
    cids = list(range(a.first, a.first + a.clouds))
    t0 = time.time()
    refs = PC.reference_fits(cids, N, K, a.na, a.nb, workers=workers)
    t_cpu = time.time() - t0
    tie_window = a.tie_window_ulps * 2.0 ** -27
    solver = PoseSolver(K, 0.1, a.na, a.nb, "cuda:0", lm_schedule="throughput", tie_window=tie_window)
    rows = []
    own = {"A": 0.0, "B": 0.0, "A_flipped": 0.0, "B_flipped": 0.0, "n": 0}
    rep = dict(A_fits=0, A_ref=0, A_hip=0, B_fits=0, B_ref=0, B_hip=0, B_hyp=0)
    t0 = time.time()
    for s in range(0, len(cids), 32):
        chunk = cids[s:s + 32]
        cl = [PC.problem(c, N, K) for c in chunk]
        DA, DB = [], []
        for c, (_cloud, p) in zip(chunk, cl):
            counts = np.bincount(np.argmax(p["instance_per_point"], 1), minlength=K)
            da, db = PC.replay_draws(100 + c, counts, a.na, a.nb)
            DA.append(da)
            DB.append(db)
        st = lambda key, which: np.stack([x[which][key] for x in cl])
        sol = solver.solve(st("P", 0), st("nocs_per_point", 1), st("instance_per_point", 1), st("joint_axis_per_point", 1),
                           st("joint_cls_gt", 1), np.stack(DA), np.stack(DB))
        sol = {k: sol[k].cpu().numpy() for k in ("baseline", "nonlinear", "best_a", "best_b", "score_b", "inliers_a", "inliers_b", "off",
                                                  "tie_a", "tie_b")}
        for b in range(len(chunk)):
            ill = PC.ill_keys(sol, b, refs[s + b], K, DA[b], DB[b])
            crows = PC.compare_cloud(sol, b, refs[s + b], K)
            # the reference's estimators on the HIP path's own masks: every flipped fit, and every 8th cloud as a control
            check_own = any(PC.flipped(r) for r in crows) or (s + b) % 8 == 0
            om = PC.own_mask_refit(sol, b, cl[b][0], cl[b][1], K) if check_own else {}
            for r in crows:
                r["cloud"] = chunk[b]
                r["ill"] = (r["stage"], r["part"]) in ill
                q = max(r["part"], 1) - 1
                r["tie"] = tuple(int(x) for x in (sol["tie_a"][b, r["part"]] if r["stage"] == "A" else sol["tie_b"][b, q]))
                if (r["stage"], r["part"]) in om:
                    r["own_mask_err"] = om[(r["stage"], r["part"])]
                    own[r["stage"]] = max(own[r["stage"]], r["own_mask_err"])
                    if PC.flipped(r):
                        own[r["stage"] + "_flipped"] = max(own[r["stage"] + "_flipped"], r["own_mask_err"])
                    own["n"] += 1
                rows.append(r)
            # winners that come from a 3-point sample with a repeated index (implementation-defined in the reference itself,
            # oracle/pose_compare.py::repeated_index): how often does one win?
            for j in range(K):
                rep["A_fits"] += 1
                rep["A_ref"] += PC.repeated_index(DA[b][j, int(refs[s + b]["iter_a"][j])])
                rep["A_hip"] += PC.repeated_index(DA[b][j, int(sol["best_a"][b, j, 0])])
            for q in range(K - 1):
                rep["B_fits"] += 1
                ir, ih = int(refs[s + b]["iter_b"][q]), int(sol["best_b"][b, q])
                rep["B_ref"] += PC.repeated_index(DB[b][q, ir, :3]) or PC.repeated_index(DB[b][q, ir, 3:])
                rep["B_hip"] += PC.repeated_index(DB[b][q, ih, :3]) or PC.repeated_index(DB[b][q, ih, 3:])
                rep["B_hyp"] += sum(PC.repeated_index(DB[b][q, i, :3]) or PC.repeated_index(DB[b][q, i, 3:]) for i in range(a.nb))
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    summ = PC.summarise(rows)
    lines = ["pose fit: HIP path vs oracle/pose_oracle.py (the reference's numpy / scipy calls) on replayed draws",
             "clouds %d (ids %d..%d), K = %d parts, N = %d points, budgets %d hypotheses per part / %d per joint, threshold 0.1"
             % (len(cids), cids[0], cids[-1], K, N, a.na, a.nb),
             "oracle: %.0f s on %d CPU workers; HIP (incl. host-side problem generation): %.0f s" % (t_cpu, workers, t_gpu), ""]
    for st_, name in (("A", "stage A  per-part RANSAC + Kabsch refit (evaluation/parallel_ancsh_pose.py:35-54)"),
                      ("B", "stage B  joint RANSAC + LM + refit (:106-194); part 0 reported from joint 1")):
        d = summ[st_]
        lines += [name,
                  "  fits %d   different winning iteration %d   different consensus set (promotion or borderline points in the winner's mask) %d = %.3f %%"
                  % (d["fits"], d["promoted"], d["different_set"], 100 * d["rate"]),
                  "  all fits: max |score difference| %.3f inliers, max inlier-mask difference %d points" % (d["max_dscore"], d["max_mask_diff"]),
                  "  same consensus set:      max |dR|, |ds|, |dt| %.3e" % d["same_set_max"],
                  "  different consensus set: max |dR| %.3e  |ds| %.3e  |dt| %.3e"
                  % (d["different_set_max_dR"], d["different_set_max_ds"], d["different_set_max_dt"]), ""]
    lines += ["3-point samples with a repeated index (rank-deficient: the reference's rotation is LAPACK's completion of a rounding-noise null space)",
              "  stage A: winner from such a sample in %d (reference arithmetic) / %d (HIP) of %d per-part fits"
              % (rep["A_ref"], rep["A_hip"], rep["A_fits"]),
              "  stage B: %.2f %% of the joint hypotheses draw one; winner from such a hypothesis in %d (reference arithmetic) / %d (HIP) of %d joint fits"
              % (100.0 * rep["B_hyp"] / max(1, rep["B_fits"] * a.nb), rep["B_ref"], rep["B_hip"], rep["B_fits"]), ""]
    lines += ["the reference's estimators (single_ / joint_transformation_estimator, :35-46,106-184) run on the HIP path's OWN winning masks vs the HIP models",
              "  (every fit of a cloud with a flipped fit + every 8th cloud: %d fits): max |dR|, |ds|, |dt|  stage A %.3e (flipped fits %.3e)   stage B %.3e (flipped fits %.3e)"
              % (own["n"], own["A"], own["A_flipped"], own["B"], own["B_flipped"]), ""]
    for st_ in ("A", "B"):
        rs = [r for r in rows if r["stage"] == st_]
        fl = [r for r in rs if PC.flipped(r)]
        flag_b = lambda r: r["tie"][0] > 0
        flag_n = lambda r: r["tie"][1] != 0
        flag_w = lambda r: r["tie"][1] < 0              # stage A: the winner's own sample is degenerate
        lines += ["the solver's own tie counts (tie_%s: [0] points within +-%.1f ulp of float32(0.1) = %.2e of the threshold under the winner, [1] degenerate contenders), stage %s:"
                  % (st_.lower(), a.tie_window_ulps, tie_window, st_),
                  "  fits with borderline points under the winner: %d of %d (%.1f %%); with a degenerate contender (repeated-index sample within one inlier of the winner): %d (%.1f %%)"
                  % (sum(map(flag_b, rs)), len(rs), 100.0 * sum(map(flag_b, rs)) / max(1, len(rs)), sum(map(flag_n, rs)), 100.0 * sum(map(flag_n, rs)) / max(1, len(rs))),
                  "  of the %d fits that ended on a different consensus set: borderline > 0 in %d, degenerate contender in %d, either in %d; neither: %d"
                  % (len(fl), sum(map(flag_b, fl)), sum(map(flag_n, fl)), sum(flag_b(r) or flag_n(r) for r in fl), sum(not (flag_b(r) or flag_n(r)) for r in fl)),
                  "  of the %d fits WITHOUT a degenerate contender: %d ended on a different consensus set"
                  % (sum(not flag_n(r) for r in rs), sum(1 for r in fl if not flag_n(r))),
                  "  same-winner mask flips (different mask, same winning iteration): %d, of which borderline > 0: %d"
                  % (sum(1 for r in fl if not r["promoted"]), sum(1 for r in fl if not r["promoted"] and flag_b(r)))]
        if st_ == "A":
            lines += ["  NEGATIVE count (the winner's own sample is degenerate; stage A, round 6): %d fits (%.2f %%), %d of them on a different consensus set "
                      "(precision %s); they hold %d of the %d flips" % (sum(map(flag_w, rs)), 100.0 * sum(map(flag_w, rs)) / max(1, len(rs)), sum(map(flag_w, fl)),
                                                                         "%d/%d" % (sum(map(flag_w, fl)), sum(map(flag_w, rs))), sum(map(flag_w, fl)), len(fl))]
        il = [r for r in rs if r["ill"]]
        lines += ["  fits whose winner (either side) comes from a repeated-index sample: %d; of these on a different consensus set: %d; their max |dR| %.3e |ds| %.3e |dt| %.3e"
                  % (len(il), sum(PC.flipped(r) for r in il), max([r["dR"] for r in il], default=0.0), max([r["ds"] for r in il], default=0.0),
                     max([r["dt"] for r in il], default=0.0)), ""]
    lines.append("largest deviations:")
    for r in sorted(rows, key=lambda r: -max(r["dR"], r["ds"], r["dt"]))[:12]:
        lines.append("  cloud %d stage %s part %d: winner %s, dscore %.2f, masks differ in %d of %d points (%d inliers): dR %.2e  ds %.2e  dt %.2e"
                     % (r["cloud"], r["stage"], r["part"], "promoted" if r["promoted"] else "same", r["dscore"], r.get("mask_diff", -1),
                        r.get("n_part", -1), r.get("n_inl", -1), r["dR"], r["ds"], r["dt"]))
    text = "\n".join(lines) + "\n"
    print(text)
    print(json.dumps(summ))
    if a.out:
        with open(os.path.splitext(a.out)[0] + "_rows.json", "w") as f:
            json.dump([{k: (list(v) if isinstance(v, tuple) else (bool(v) if isinstance(v, (bool, np.bool_)) else v)) for k, v in r.items()} for r in rows], f)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
