"""Tie promotions, measured instead of skipped: the HIP pose fit against oracle/pose_oracle.py (the reference's numpy / scipy calls,
evaluation/parallel_ancsh_pose.py:20-54,106-194) on replayed draws over a few hundred part fits.  Where both paths end on the same
consensus set (same winning hypothesis, identical inlier mask: 99 % of the fits) R, s, t agree to 1e-5.  Where they do not (float32
residuals within one rounding of the 0.1 threshold counted on one side only) the scores differ by at most one inlier, the masks by
a handful of points, and the final refits stay within the bounds measured in profiles/r05_pose_tie_rate_full.txt; the RATE of such
fits is bounded too (oracle/pose_compare.py holds the bars and says where each number comes from)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.pose_compare import FLIPPED_RATE_MAX  # noqa: E402   (measured: profiles/r04_pose_tie_rate.txt)


def _solve(dev, cids, K, N, na, nb):
    from articulated_pose_amd.pose import PoseSolver
    from articulated_pose_amd.pose.parallel_ancsh_pose import draws_from_seed
    from oracle import pose_compare as PC
    cl = [PC.problem(c, N, K) for c in cids]
    DA, DB = [], []
    for c, (_cloud, p) in zip(cids, cl):
        counts = np.bincount(np.argmax(p["instance_per_point"], 1), minlength=K)
        da, db = PC.replay_draws(100 + c, counts, na, nb)
        if na <= 2000:
            pa, pb = draws_from_seed(100 + c, counts, na, nb)
            assert np.array_equal(da, pa) and np.array_equal(db, pb)          # the checker's stream == the product's
        DA.append(da)
        DB.append(db)
    st = lambda key, which: np.stack([x[which][key] for x in cl])
    sol = PoseSolver(K, 0.1, na, nb, dev, lm_schedule="throughput").solve(
        st("P", 0), st("nocs_per_point", 1), st("instance_per_point", 1), st("joint_axis_per_point", 1), st("joint_cls_gt", 1),
        np.stack(DA), np.stack(DB))
    keys = ("baseline", "nonlinear", "best_a", "best_b", "score_b", "inliers_a", "inliers_b", "off", "tie_a", "tie_b", "record")
    return {k: sol[k].cpu().numpy() for k in keys}, cl, DA, DB


# (K, N, clouds, hypotheses per part, per joint): round 4's sample at the reduced budget, and -- round 5 -- the REFERENCE'S budgets
# (evaluation/parallel_ancsh_pose.py:262,299: niter = 10000 / 200) on >= 8 clouds per configuration
@pytest.mark.parametrize("K,N,n_clouds,na,nb", [(3, 1024, 48, 2000, 64), (4, 2048, 12, 2000, 64), (2, 2048, 12, 2000, 64),
                                                (3, 1024, 16, 10000, 200), (4, 2048, 8, 10000, 200), (2, 2048, 8, 10000, 200)])
def test_tie_promotions_are_rare_and_bounded(dev, oracle, K, N, n_clouds, na, nb):
    from oracle import cpu_layout, pose_compare as PC
    cids = list(range(9000, 9000 + n_clouds))
    refs = PC.reference_fits(cids, N, K, na, nb, workers=max(1, min(12, cpu_layout.usable_cpus() - 2)))
    sol, cl, DA, DB = _solve(dev, cids, K, N, na, nb)
    rows = [dict(r, cloud=b) for b in range(n_clouds) for r in PC.compare_cloud(sol, b, refs[b], K, draws=(DA[b], DB[b]), problem_data=cl[b])]
    fits, different = PC.check_rows(rows)          # same set: 1e-5 / 1e-4; different set: its bounds (looser for repeated-index winners) + the own-mask refit
    assert fits == 2 * K * n_clouds
    # the rate over this sample must be compatible with the measured one (slack for the small sample)
    assert different <= max(3, int(np.ceil(2 * FLIPPED_RATE_MAX * fits))), (different, fits)
    # the pose record the finish kernels write IS [baseline | nonlinear]
    assert np.array_equal(sol["record"][:, :, :13], sol["baseline"], equal_nan=True) and np.array_equal(sol["record"][:, :, 13:], sol["nonlinear"], equal_nan=True)
    # at the reference's budgets every fit that ended on another consensus set had a repeated-index winner on one side (32 of 32 in
    # profiles/r05_pose_tie_rate_full.txt) and no winner had a point within 32 ulp of the threshold (tie[..., 0] == 0 in 8064 fits)
    if na >= 10000:
        regular_different = sum(1 for r in rows if PC.flipped(r) and not r["ill"])
        assert regular_different <= 1, [r for r in rows if PC.flipped(r) and not r["ill"]]
        assert int((sol["tie_a"][:, :, 0] > 0).sum()) + int((sol["tie_b"][:, :, 0] > 0).sum()) <= max(1, fits // 50)
    # tie[..., 1] counts the degenerate contenders of THIS implementation's arithmetic: a fit whose own winner comes from a repeated-index
    # sample must carry it
    for r in rows:
        q = max(r["part"], 1) - 1
        it = int(sol["best_a"][r["cloud"], r["part"], 0] if r["stage"] == "A" else sol["best_b"][r["cloud"], q])
        own_ill = PC.repeated_index(DA[r["cloud"]][r["part"], it]) if r["stage"] == "A" else (
            PC.repeated_index(DB[r["cloud"]][q, it, :3]) or PC.repeated_index(DB[r["cloud"]][q, it, 3:]))
        if r["stage"] == "A":       # round 6: the SIGN of stage A's count says whether the winner's own sample is degenerate -- exactly then
            assert (int(sol["tie_a"][r["cloud"], r["part"], 1]) < 0) == bool(own_ill), r
        elif own_ill:
            assert int(sol["tie_b"][r["cloud"], q, 1]) >= 1, r
