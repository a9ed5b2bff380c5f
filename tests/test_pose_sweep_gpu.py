"""GPU: seeded sweep of the whole pose fit (stage A RANSAC + refit, stage B joint LM fits + refit) over ragged problems -- K = 2 / 3 / 4,
revolute and prismatic clouds of 96..700 points, skewed part sizes (one part may hold a few dozen points), noisier predictions than
the fixed tests use, odd budgets -- each cloud solved by the HIP path and by oracle/pose_oracle.py (the reference's numpy / scipy
calls) on REPLAYED draws and compared fit by fit with the bars of oracle/pose_compare.py (same consensus set: 1e-5 / 1e-4; a fit that
ends on another consensus set stays inside the measured bounds -- its own, looser one when a winner comes from a repeated-index
sample -- and is counted)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# 24 / 16 seeds in the suite; ANCSH_POSE_SWEEP_SEEDS=N for a one-off long fuzz (profiles/r05_ops_fuzz.txt)
import os
POSE_SEEDS = int(os.environ.get("ANCSH_POSE_SWEEP_SEEDS", "0"))


def _problem(seed):
    from articulated_pose_amd.synthetic import make_cloud, make_predictions
    rng = np.random.RandomState(seed)
    K = int(rng.choice([2, 3, 4]))
    N = int(rng.choice([96, 160, 257, 400, 512, 700]))
    jt = "prismatic" if (K == 4 and seed % 2) else "revolute"
    c = make_cloud(7000 + seed, N=N, K=K, joint_type=jt)
    p = make_predictions(c, K, seed=seed, noise=float(rng.choice([0.005, 0.01, 0.02])), outlier=float(rng.choice([0.0, 0.1, 0.25])),
                         flip=float(rng.choice([0.0, 0.05, 0.15])))
    if seed % 3 == 0:            # squeeze one part: relabel most of its predicted points to a neighbour (a few dozen stay)
        W = p["instance_per_point"].copy()
        lab = np.argmax(W, 1)
        j = int(rng.randint(1, K))
        idx = np.nonzero(lab == j)[0]
        move = idx[24:] if len(idx) > 24 else idx[:0]
        W[move] = W[move][:, np.roll(np.arange(K), 1)]
        p["instance_per_point"] = W
    na, nb = int(rng.choice([33, 64, 150])), int(rng.choice([4, 8, 17]))
    return c, p, K, na, nb


@pytest.mark.parametrize("seed", range(POSE_SEEDS or 24))
def test_pose_fit_sweep(dev, seed):
    from articulated_pose_amd.pose import PoseSolver
    from articulated_pose_amd.pose.parallel_ancsh_pose import draws_from_seed
    from oracle import pose_compare as PC, pose_oracle as PO
    c, p, K, na, nb = _problem(seed)
    counts = np.bincount(np.argmax(p["instance_per_point"], 1), minlength=K)
    if counts.min() < 3:
        pytest.skip("a part with fewer than three predicted points: the reference raises")
    da, db = draws_from_seed(500 + seed, counts, na, nb)
    ref = PO.solve_cloud(c["P"], p["nocs_per_point"], p["instance_per_point"], p["joint_axis_per_point"], p["joint_cls_gt"], K,
                         [PO.SampleStream(list(da[j])) for j in range(K)],
                         [PO.SampleStream([d for row in db[j] for d in (row[:3], row[3:])]) for j in range(K - 1)], 0.1, na, nb)
    sol = PoseSolver(K, 0.1, na, nb, dev).solve(c["P"][None], p["nocs_per_point"][None], p["instance_per_point"][None],
                                                p["joint_axis_per_point"][None], p["joint_cls_gt"][None], da[None], db[None])
    s_np = {k: sol[k].cpu().numpy() for k in ("baseline", "nonlinear", "best_a", "best_b", "score_b", "inliers_a", "inliers_b", "off")}
    packed = PC.pack(ref, K)
    rows = PC.compare_cloud(s_np, 0, packed, K, draws=(da, db), problem_data=(c, p))
    # A winner -- here or in the oracle -- that comes from a 3-point sample with a repeated index is implementation-defined in the
    # reference itself (oracle/pose_compare.py::repeated_index: LAPACK's completion of a rounding-noise null space seeds the LM fit).
    # Round 4 exempted those fits; since round 5 they are HELD to their own measured bound (ILL_BOUNDS, profiles/r05_pose_tie_rate_full.txt)
    # and every fit that ends on another consensus set must still be the reference's refit of the set it ended on (own_mask_err).
    # With parts of 24 points and 4 hypotheses per joint a repeated-index winner is common; at the reference's budgets it is 1 % of the fits.
    fits, different = PC.check_rows(rows, ill_value_bars=not POSE_SEEDS)
    regular_different = sum(1 for r in rows if PC.flipped(r) and not r["ill"])
    assert fits == 2 * K and regular_different <= 1, (fits, different, regular_different)
    # the solver's own flag: a fit without degenerate contenders ends on the reference arithmetic's consensus set
    tie_a, tie_b = sol["tie_a"].cpu().numpy()[0], sol["tie_b"].cpu().numpy()[0]
    for r in rows:
        if r["ill"] and int(s_np["best_a"][0, r["part"], 0] if r["stage"] == "A" else s_np["best_b"][0, max(r["part"], 1) - 1]) >= 0:
            q = max(r["part"], 1) - 1
            own_winner_ill = (PC.repeated_index(da[r["part"], int(s_np["best_a"][0, r["part"], 0])]) if r["stage"] == "A" else
                              PC.repeated_index(db[q, int(s_np["best_b"][0, q]), :3]) or PC.repeated_index(db[q, int(s_np["best_b"][0, q]), 3:]))
            if own_winner_ill:          # stage A reports its own degenerate winner with a NEGATIVE count (round 6), stage B with a count >= 1
                assert (tie_a[r["part"], 1] <= -1) if r["stage"] == "A" else (tie_b[q, 1] >= 1), r
    np.testing.assert_array_equal(sol["counts"].cpu().numpy()[0], counts)


@pytest.mark.filterwarnings("ignore:.*encountered in scalar")      # the reference arithmetic divides by a zero variance on a 5-sample of one point
@pytest.mark.parametrize("seed", range(POSE_SEEDS or 16))
def test_umeyama_and_similarity_ransac_sweep(dev, seed):
    """ancsh_umeyama and ancsh_estimate_similarity_transform (lib/aligning.py:580-622, :17-32 and :485-547) on a ragged batch of seeded
    problems -- 3..3000 points, scales 0.2..5, noise 0..0.05, outlier shares 0..40 % -- against oracle/pose_oracle.py's restatement of
    the same numpy calls on the float32-cast points, draws replayed; one launch per call for the whole batch."""
    from articulated_pose_amd.pose import estimate_similarity_transform_batch, umeyama_batch
    from oracle import pose_oracle as PO
    rng = np.random.RandomState(9000 + seed)
    srcs, tgts, draws = [], [], []
    for k in range(5):
        n = int(rng.choice([3, 5, 17, 100, 341, 1024, 3000])) if k else 5 + seed
        q, _ = np.linalg.qr(rng.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        s, t = rng.uniform(0.2, 5.0), rng.randn(3)
        src = (rng.rand(n, 3) - 0.5).astype(np.float32)
        tgt = (s * src @ q.T + t + rng.randn(n, 3) * float(rng.choice([0.0, 0.005, 0.05]))).astype(np.float32)
        bad = rng.rand(n) < float(rng.choice([0.0, 0.1, 0.4]))
        tgt[bad] = (rng.rand(int(bad.sum()), 3) * 4 - 2).astype(np.float32)
        srcs.append(src); tgts.append(tgt)
        d = rng.randint(n, size=(100, 5))
        # A 5-sample with fewer than THREE distinct points has a covariance of rank <= 1: its rotation is LAPACK's completion of a null
        # space in the reference (np.linalg.svd, lib/aligning.py:592) and the shortest-arc member of the optimal family here --
        # implementation-defined in the reference itself, and since every later step depends on which points that arbitrary model
        # happens to pass (PassThreshold is of the size of the object), so is everything after it.  The long fuzz (ANCSH_POSE_SWEEP_SEEDS
        # = 300, profiles/r05_ops_fuzz.txt) met three such cases, all on 3-point problems.  The replayed streams therefore redraw such
        # rows; test_degenerate_five_samples below feeds them and asks for a well-formed answer only.
        for i in range(len(d)):
            while len(set(d[i].tolist())) < 3:
                d[i] = rng.randint(n, size=5)
        draws.append(d)
    got = umeyama_batch(srcs, tgts, dev)
    for k in range(5):
        hom = lambda a: np.vstack([a.T.astype(np.float64), np.ones((1, a.shape[0]))])
        S, R, T, Out = PO.estimateSimilarityUmeyama(hom(srcs[k]), hom(tgts[k]))
        for g, w in zip(got[k], (S, R, T, Out)):
            np.testing.assert_allclose(g, w, rtol=0, atol=1e-9 * max(1.0, float(np.abs(w).max())), err_msg="umeyama problem %d (n = %d)" % (k, len(srcs[k])))
    got = estimate_similarity_transform_batch(srcs, tgts, draws=np.stack(draws), device=dev)
    for k in range(5):
        want = PO.estimateSimilarityTransform(srcs[k].astype(np.float64), tgts[k].astype(np.float64), draws[k])
        if want[0] is None:
            assert got[k] == (None, None, None, None), k
            continue
        assert got[k][0] is not None, k
        for g, w in zip(got[k], want):
            np.testing.assert_allclose(g, w, rtol=0, atol=1e-8 * max(1.0, float(np.abs(w).max())), err_msg="ransac problem %d (n = %d)" % (k, len(srcs[k])))


def test_degenerate_five_samples(dev):
    """estimateSimilarityTransform on 5-samples that repeat ONE or TWO points (rank <= 1 covariances; np.random.randint draws with
    replacement, so 3-point parts meet them all the time): the reference's answer depends on LAPACK's null-space completion, so no value
    is compared -- the call must come back with either four Nones or a finite similarity whose rotation is orthonormal."""
    from articulated_pose_amd.pose import estimate_similarity_transform_batch
    rng = np.random.RandomState(5)
    srcs, tgts, draws = [], [], []
    for k, n in enumerate([3, 3, 4, 5, 17]):
        src = (rng.rand(n, 3) - 0.5).astype(np.float32)
        q, _ = np.linalg.qr(rng.randn(3, 3))
        tgt = (2.0 * src @ q.T + rng.randn(3)).astype(np.float32)
        d = rng.randint(n, size=(100, 5))
        d[:50] = d[:50, :1]                                   # one point five times
        d[50:80, 2:] = d[50:80, 1:2]                          # two distinct points
        srcs.append(src); tgts.append(tgt); draws.append(d)
    got = estimate_similarity_transform_batch(srcs, tgts, draws=np.stack(draws), device=dev)
    for k in range(5):
        if got[k][0] is None:
            assert got[k] == (None, None, None, None)
            continue
        S, R, T, Out = got[k]
        assert np.isfinite(S).all() and np.isfinite(R).all() and np.isfinite(T).all() and np.isfinite(Out).all(), k
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-9, err_msg=str(k))
