#!/usr/bin/env python
"""The pose-fit sweep of tests/test_pose_sweep_gpu.py over MANY seeds, summarised by category instead of asserted row by row:
    python tools/pose_fuzz.py [seeds=800]
Problems: K = 2 / 3 / 4, 96..700 points, parts squeezed to a few dozen points, noise 0.005..0.02, outliers 0..25 %, label flips 0..15 %,
33..150 hypotheses per part and 4..17 per joint -- deliberately ill-posed.  HIP path vs oracle/pose_oracle.py on replayed draws,
lined up by oracle/pose_compare.py.  profiles/r05_ops_fuzz.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    import articulated_pose_amd  # noqa: F401
    from articulated_pose_amd.pose import PoseSolver
    from articulated_pose_amd.pose.parallel_ancsh_pose import draws_from_seed
    from oracle import pose_compare as PC, pose_oracle as PO
    import test_pose_sweep_gpu as TS
    rows, skipped, solvers = [], 0, {}
    for seed in range(n):
        c, p, K, na, nb = TS._problem(seed)
        counts = np.bincount(np.argmax(p["instance_per_point"], 1), minlength=K)
        if counts.min() < 3:
            skipped += 1
            continue
        da, db = draws_from_seed(500 + seed, counts, na, nb)
        ref = PO.solve_cloud(c["P"], p["nocs_per_point"], p["instance_per_point"], p["joint_axis_per_point"], p["joint_cls_gt"], K,
                             [PO.SampleStream(list(da[j])) for j in range(K)],
                             [PO.SampleStream([d for row in db[j] for d in (row[:3], row[3:])]) for j in range(K - 1)], 0.1, na, nb)
        sv = solvers.setdefault((K, na, nb), PoseSolver(K, 0.1, na, nb, "cuda:0"))
        sol = sv.solve(c["P"][None], p["nocs_per_point"][None], p["instance_per_point"][None], p["joint_axis_per_point"][None],
                       p["joint_cls_gt"][None], da[None], db[None])
        s_np = {k: sol[k].cpu().numpy() for k in ("baseline", "nonlinear", "best_a", "best_b", "score_b", "inliers_a", "inliers_b", "off")}
        for r in PC.compare_cloud(s_np, 0, PC.pack(ref, K), K, draws=(da, db), problem_data=(c, p)):
            r["seed"] = seed
            rows.append(r)
    print("# tools/pose_fuzz.py %d: %d problems (%d skipped: a part with fewer than three predicted points, the reference raises), %d reported fits"
          % (n, n - skipped, skipped, len(rows)))
    for st in ("A", "B"):
        rs = [r for r in rows if r["stage"] == st]
        th = [r for r in rs if PC.thin(r)]
        reg = [r for r in rs if not PC.thin(r)]
        same = [r for r in reg if not PC.flipped(r)]
        fl = [r for r in reg if PC.flipped(r)]
        fl_ill = [r for r in fl if r["ill"]]
        fl_reg = [r for r in fl if not r["ill"]]
        mx = lambda L: max([max(r["dR"], r["ds"], r["dt"]) for r in L], default=0.0)
        print("stage %s: %d fits | thin consensus (< 3 inliers of a part; counted, not compared): %d (max deviation %.3g)" % (st, len(rs), len(th), mx(th)))
        print("   same consensus set: %d fits, max |dR|,|ds|,|dt| %.3g (bar %g)" % (len(same), mx(same), PC.TOL_SAME_SET if st == "A" else PC.TOL_SAME_SET_B))
        print("   different set, a repeated-index winner: %d (max deviation %.3g, max own-mask refit error %.3g)"
              % (len(fl_ill), mx(fl_ill), max([r.get("own_mask_err", 0.0) for r in fl_ill], default=0.0)))
        print("   different set, regular winners: %d (max deviation %.3g, max dscore %.2f, max own-mask refit error %.3g)"
              % (len(fl_reg), mx(fl_reg), max([r["dscore"] for r in fl_reg], default=0.0), max([r.get("own_mask_err", 0.0) for r in fl_reg], default=0.0)))
        worst = sorted(same, key=lambda r: -max(r["dR"], r["ds"], r["dt"]))[:3]
        for r in worst:
            print("      worst same-set: seed %d part %d  dR %.3g ds %.3g dt %.3g  inliers %d of %d  ill %s" % (r["seed"], r["part"], r["dR"], r["ds"], r["dt"], r["n_inl"], r["n_part"], r["ill"]))


if __name__ == "__main__":
    main()
