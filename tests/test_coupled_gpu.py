"""The PRODUCTION data flow: AncshPipeline(couple=True) -- the pose fit consumes the networks' own outputs (NOCS and part
mask from the NPCS network, joint axes from the ANCSH network: evaluation/parallel_ancsh_pose.py:232-237,295), inside
the captured hipGraph, with several batches in flight.

(i)  hand-built weights (tests/helpers.py::passthrough_pose_problem) make the heads emit a usable segmentation and
     part-NOCS, so the coupled path has a known answer AND can be compared with the CPU oracle run end to end
     (net_oracle -> pose_oracle) on replayed numpy sample streams;
(ii) seeded random weights give near-uniform masks, i.e. degenerate / empty parts: the path must stay deterministic,
     finish, and report empty parts as NaN rows with best_iter -1 (the reference raises inside randint(0) there)."""
import numpy as np
import pytest
import torch

from helpers import passthrough_pose_problem

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _bits(t):
    return t.contiguous().view(torch.int64)          # NaN-safe exact comparison of float64 records


@pytest.mark.parametrize("K", [3, 2])
def test_coupled_pipeline_matches_end_to_end_oracle(dev, K):
    from articulated_pose_amd.pipeline import AncshPipeline
    from articulated_pose_amd.pose.d3_utils import rot_diff_degree
    from articulated_pose_amd.pose.parallel_ancsh_pose import draws_from_seed
    from oracle import net_oracle
    from oracle import pose_oracle as PO
    B, N, na, nb = 3, 1024, 400, 32
    pb = passthrough_pose_problem(K, B, N, seed=K)
    # CPU oracle, end to end: both networks, then the reference's per-cloud solve on their outputs
    on = net_oracle.forward(pb["w_npcs"], pb["P"], K, mixed_pred=False, early_split_nocs=False)
    oa = net_oracle.forward(pb["w_ancsh"], pb["P"], K)
    lab = on["W"].argmax(2)
    assert np.array_equal(lab, pb["cls"])             # the hand-built segmentation head separates the slabs
    DA, DB, want = [], [], []
    for b in range(B):
        counts = np.bincount(lab[b], minlength=K)
        da, db = draws_from_seed(300 + b, counts, na, nb)
        DA.append(da); DB.append(db)
        sa = [PO.SampleStream(list(da[j])) for j in range(K)]
        sb = [PO.SampleStream([d for row in db[j] for d in (row[:3], row[3:])]) for j in range(K - 1)]
        want.append(PO.solve_cloud(pb["P"][b], on["nocs_per_point"][b], on["W"][b], oa["joint_axis_per_point"][b], pb["cls"][b],
                                   K, sa, sb, 0.1, na, nb))
    recs = []
    for use_graph, slots in ((False, 1), (True, 2)):
        pipe = AncshPipeline(K, pb["w_ancsh"], pb["w_npcs"], B, N, dev, couple=True, use_graph=use_graph, slots=slots,
                             niter_a=na, niter_b=nb)
        pipe.load_inputs(pb["P"], pb["cls"])
        pipe.load_draws(np.stack(DA), np.stack(DB))
        pipe.prepare()
        for _ in range(slots + 1):
            sl, out = pipe.step()
            sl.stream.synchronize()
            recs.append(out["record"].clone())
            assert torch.equal(out["pose"]["labels"].cpu(), torch.from_numpy(lab.astype(np.int32)))   # integer labels: exact
    assert all(torch.equal(_bits(r), _bits(recs[0])) for r in recs[1:])       # eager == graph == every slot / replay
    rec = recs[0].cpu().numpy()                                                # (B, K, 26) = [baseline 13 | nonlinear 13]
    for b in range(B):
        for j in range(K):
            for kind, o in (("baseline", 0), ("nonlinear", 13)):
                R, s, t = want[b][kind][j]
                np.testing.assert_allclose(rec[b, j, o:o + 9].reshape(3, 3), np.asarray(R, np.float64), atol=TOL)
                np.testing.assert_allclose(rec[b, j, o + 9], float(s), atol=TOL * max(1.0, float(s)))
                np.testing.assert_allclose(rec[b, j, o + 10:o + 13], np.asarray(t, np.float64), atol=TOL * max(1.0, float(s)))
            # and the fit is the pose the heads were built from (sigmoid's cubic term costs ~1e-3 of a NOCS unit)
            assert rot_diff_degree(rec[b, j, 13:22].reshape(3, 3), pb["R"][j]) < 1.0
            assert abs(rec[b, j, 22] / pb["s"][j] - 1) < 0.02


def test_coupled_pipeline_with_random_weights_is_deterministic_and_reports_empty_parts(dev):
    from articulated_pose_amd.pipeline import AncshPipeline
    from articulated_pose_amd.synthetic import make_batch
    from articulated_pose_amd.weights import synthetic_weights
    K, B, N = 3, 8, 1024
    wa = synthetic_weights(K, seed=0)
    wn = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1)
    batch = make_batch(500, B, N=N, K=K)
    outs = []
    for use_graph, slots in ((False, 1), (True, 3)):
        pipe = AncshPipeline(K, wa, wn, B, N, dev, couple=True, use_graph=use_graph, slots=slots, seed=11, niter_a=500, niter_b=40)
        pipe.load_inputs(batch["P"], batch["cls_gt"])
        pipe.prepare()
        for _ in range(slots + 1):
            sl, out = pipe.step()
            sl.stream.synchronize()                                           # returns: no lane spins on a degenerate part
            outs.append((out["record"].clone(), out["pose"]["counts"].clone(), out["pose"]["best_a"].clone()))
    rec0, cnt0, best0 = outs[0]
    for rec, cnt, best in outs[1:]:
        assert torch.equal(_bits(rec), _bits(rec0)) and torch.equal(cnt, cnt0) and torch.equal(best, best0)
    rec, cnt, best = rec0.cpu().numpy(), cnt0.cpu().numpy(), best0.cpu().numpy()
    assert cnt.sum(1).tolist() == [N] * B                                     # every point belongs to exactly one predicted part
    for b in range(B):
        for j in range(K):
            if cnt[b, j] == 0:                                                # empty part: NaN row, winning iteration -1
                assert np.isnan(rec[b, j, :13]).all() and best[b, j, 0] == -1
            else:
                assert best[b, j, 0] >= 0 and 0 <= best[b, j, 1] <= cnt[b, j]
                if cnt[b, j] >= 3:
                    assert np.isfinite(rec[b, j, :13]).all()


def test_coupled_pipeline_part_never_predicted(dev):
    """The NPCS segmentation head never predicts part 2 (bias -50): in every cloud part 2 is empty.  Stage A row 2 and the
    joint fit (0,2) must come back as NaN / -1 while parts 0, 1 and joint (0,1) are solved normally, under graph replay."""
    from articulated_pose_amd.pipeline import AncshPipeline
    from articulated_pose_amd.synthetic import make_batch
    from articulated_pose_amd.weights import synthetic_weights
    K, B, N = 3, 4, 1024
    wa = synthetic_weights(K, seed=0)
    wn = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1)
    wn["SPFN/nocs_net/fc2_0/biases"] = wn["SPFN/nocs_net/fc2_0/biases"].copy()
    wn["SPFN/nocs_net/fc2_0/biases"][2] = -50.0
    batch = make_batch(700, B, N=N, K=K)
    pipe = AncshPipeline(K, wa, wn, B, N, dev, couple=True, use_graph=True, slots=2, seed=3, niter_a=300, niter_b=24)
    pipe.load_inputs(batch["P"], batch["cls_gt"])
    pipe.prepare()
    got = []
    for _ in range(3):
        sl, out = pipe.step()
        sl.stream.synchronize()
        got.append((out["record"].clone(), out["pose"]["counts"].clone(), out["pose"]["best_a"].clone(), out["pose"]["best_b"].clone()))
    for g in got[1:]:
        assert torch.equal(_bits(g[0]), _bits(got[0][0]))
    rec, cnt, best_a, best_b = [x.cpu().numpy() for x in got[0]]
    assert (cnt[:, 2] == 0).all() and (cnt[:, :2].sum(1) == N).all()
    assert np.isnan(rec[:, 2, :]).all() and (best_a[:, 2, 0] == -1).all() and (best_b[:, 1] == -1).all()
    ok = cnt[:, :2].min(1) >= 3
    assert np.isfinite(rec[ok][:, :2, :13]).all()
