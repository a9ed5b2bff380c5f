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
    solver = PoseSolver(K, 0.1, a.na, a.nb, "cuda:0", lm_schedule="throughput")
    rows = []
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
        sol = {k: sol[k].cpu().numpy() for k in ("baseline", "nonlinear", "best_a", "best_b", "score_b", "inliers_a", "inliers_b", "off")}
        for b in range(len(chunk)):
            for r in PC.compare_cloud(sol, b, refs[s + b], K):
                r["cloud"] = chunk[b]
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
    lines.append("largest deviations:")
    for r in sorted(rows, key=lambda r: -max(r["dR"], r["ds"], r["dt"]))[:12]:
        lines.append("  cloud %d stage %s part %d: winner %s, dscore %.2f, masks differ in %d of %d points (%d inliers): dR %.2e  ds %.2e  dt %.2e"
                     % (r["cloud"], r["stage"], r["part"], "promoted" if r["promoted"] else "same", r["dscore"], r.get("mask_diff", -1),
                        r.get("n_part", -1), r.get("n_inl", -1), r["dR"], r["ds"], r["dt"]))
    text = "\n".join(lines) + "\n"
    print(text)
    print(json.dumps(summ))
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
