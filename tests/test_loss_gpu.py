"""GPU: ancsh_test_losses (through articulated_pose_amd.loss) against the CPU oracle (float tolerance 1e-5: the kernel sums in
float64, TensorFlow / the oracle in float32), and predict_and_save's test_loss.txt end to end: ragged raw clouds ->
on-GPU input sampling -> network forward -> losses, checked against the oracle pipeline on the same inputs."""
import numpy as np
import pytest
import torch

from test_loss_cpu import fake_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,K,mixed,type_l", [(3, 257, 3, True, "L2"), (2, 1024, 2, False, "L2"), (5, 64, 4, True, "L1"), (1, 2048, 8, True, "L2")])
def test_losses_match_oracle(dev, B, N, K, mixed, type_l):
    from articulated_pose_amd import loss as L
    from oracle import loss_oracle as LO
    pred, gt = fake_batch(B, N, K, seed=B + K, mixed=mixed)
    ld = L.compute_loss({k: torch.from_numpy(v).to(dev) for k, v in pred.items()}, gt, K, mixed, type_l)
    want = LO.loss_dict(pred, gt, K, mixed, type_l)
    assert set(want) == set(k for k in ld if k != "_keep")
    for k in want:
        np.testing.assert_allclose(ld[k].cpu().numpy(), want[k], rtol=1e-5, atol=1e-6, err_msg=k)
    tot, wtot = L.collect_losses(ld, mixed), LO.collect_losses(want, mixed)
    for k in wtot:
        assert abs(tot[k] - wtot[k]) <= 1e-5 * max(1.0, abs(wtot[k])), k
    with pytest.raises(ValueError):
        L.compute_loss({k: torch.from_numpy(v).to(dev) for k, v in pred.items()}, gt, K + 1, mixed, type_l)
    with pytest.raises(ValueError):
        L.compute_loss({k: torch.from_numpy(v).to(dev) for k, v in pred.items()}, gt, K, mixed, "Soft_L1")


# 12 seeds in the suite; ANCSH_LOSS_SWEEP_SEEDS=N for a one-off long fuzz (profiles/r05_ops_fuzz.txt)
import os
LOSS_SEEDS = range(int(os.environ.get("ANCSH_LOSS_SWEEP_SEEDS", "12")))


@pytest.mark.parametrize("seed", LOSS_SEEDS)
def test_losses_sweep(dev, seed):
    """Seeded sweep of ancsh_test_losses over ragged batches -- 1..6 clouds of 1..3000 points, 1..8 parts, mixed / plain heads, L1 / L2 --
    against the CPU oracle, every loss and every collected total."""
    from articulated_pose_amd import loss as L
    from oracle import loss_oracle as LO
    rng = np.random.RandomState(6000 + seed)
    B, K = int(rng.randint(1, 7)), int(rng.randint(1, 9))
    N = int([1, 2, 3, 31, 64, 65, 257][rng.randint(7)]) if seed % 3 == 0 else int(rng.randint(1, 3001))
    mixed, type_l = bool(rng.randint(2)), ["L1", "L2"][rng.randint(2)]
    pred, gt = fake_batch(B, N, K, seed=100 + seed, mixed=mixed)
    ld = L.compute_loss({k: torch.from_numpy(v).to(dev) for k, v in pred.items()}, gt, K, mixed, type_l)
    want = LO.loss_dict(pred, gt, K, mixed, type_l)
    assert set(want) == set(k for k in ld if k != "_keep")
    for k in want:
        np.testing.assert_allclose(ld[k].cpu().numpy(), want[k], rtol=1e-5, atol=1e-6, err_msg="%s B=%d N=%d K=%d %s %s" % (k, B, N, K, mixed, type_l))
    tot, wtot = L.collect_losses(ld, mixed), LO.collect_losses(want, mixed)
    for k in wtot:
        assert abs(tot[k] - wtot[k]) <= 1e-5 * max(1.0, abs(wtot[k])), (k, B, N, K)


@pytest.mark.parametrize("ti", range(4))
def test_losses_match_the_interpreted_reference_trace(dev, ti):
    """The HIP losses against tests/golden/loss_trace.json -- the op trace the reference's own lib/loss.py + lib/network.py leave under a
    recording stand-in -- interpreted in numpy float32 (tests/test_loss_trace_cpu.py): no self-written oracle in between."""
    from articulated_pose_amd import loss as L
    import test_loss_trace_cpu as T
    t = T.TRACES[ti]
    K, mixed = t["n_max_parts"], t["flags"]["is_mixed"]
    pred, gt = fake_batch(4, 1024, K, seed=10 + ti, mixed=mixed)
    ld = L.compute_loss({k: torch.from_numpy(v).to(dev) for k, v in pred.items()}, gt, K, mixed, t["config"]["coord_regress_loss"])
    want, wtot, _ = T.run(t, pred, gt)
    assert set(want) == set(k for k in ld if k != "_keep")
    for k in want:
        np.testing.assert_allclose(ld[k].cpu().numpy(), want[k], rtol=1e-5, atol=1e-6, err_msg=k)
    tot = L.collect_losses(ld, mixed, t["flags"]["pred_joint"], t["flags"]["pred_joint_ind"])
    assert set(tot) == set(wtot)
    for k in wtot:
        assert abs(tot[k] - float(wtot[k])) <= 1e-5 * max(1.0, abs(float(wtot[k]))), k


def test_predict_and_save_writes_test_loss_txt(dev, tmp_path):
    """raw ragged clouds -> create_unit_data_batch (GPU) -> Network.predict_and_save -> test_loss.txt; the same numbers from
    the oracle chain (input_oracle -> net_oracle -> loss_oracle)."""
    from articulated_pose_amd import loss as L
    from articulated_pose_amd.dataset import create_unit_data_batch, tiled_size
    from articulated_pose_amd.network import Network
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import input_oracle, loss_oracle as LO, net_oracle
    from golden.gen_input_golden import synthetic_parts
    K, N = 3, 1024
    rng = np.random.RandomState(12)
    clouds = [synthetic_parts(rng, s) for s in ([500, 400, 300], [90, 70, 60], [1000, 800, 700], [300, 300, 500], [64, 1500, 200])]
    nfs = [0.5, 0.6, 0.45, 0.7, 0.55]
    perms = [np.random.RandomState(40 + i).permutation(tiled_size(sum(len(a) for a in c["parts_pts"]), N)) for i, c in enumerate(clouds)]
    w = synthetic_weights(K, seed=0)
    net = Network(K, w, "ancsh", dev)
    batches, want_sum, n = [], {}, 0
    for s in (0, 2, 4):                                    # batch sizes 2, 2, 1: the running mean is weighted by the batch size
        sel = list(range(s, min(s + 2, 5)))
        b = create_unit_data_batch([clouds[i] for i in sel], N, [nfs[i] for i in sel], K, perms=[perms[i] for i in sel], device=dev)
        b["basename_list"] = ["0001_0_%d" % i for i in sel]
        batches.append(b)
        recs = [input_oracle.create_unit_data(clouds[i], N, np.float32(nfs[i]), K, perm=perms[i]) for i in sel]
        gt = {k: np.stack([r[k] for r in recs]) for k in recs[0]}
        np.testing.assert_array_equal(b["P"].cpu().numpy(), gt["P"])
        pred = net_oracle.forward(w, gt["P"], K)
        tot = LO.collect_losses(LO.loss_dict(pred, dict(gt, cls_gt=gt["cls_gt"].astype(int), joint_cls_gt=gt["joint_cls_gt"].astype(int)), K, True), True)
        for k, v in tot.items():
            want_sum[k] = want_sum.get(k, 0.0) + v * len(sel)
        n += len(sel)
    res = net.predict_and_save(batches, str(tmp_path))
    assert res["n"] == 5
    want = {k: v / n for k, v in want_sum.items()}
    for k in want:
        assert abs(res["losses"][k] - want[k]) <= 1e-4 * max(1.0, abs(want[k])), (k, res["losses"][k], want[k])
    txt = open(tmp_path / "test_loss.txt").read()
    assert txt == res["msg"] == L.format_loss_result(res["losses"], True)
    assert txt.startswith("Total Loss: ") and ", gocs Loss: " in txt and txt.endswith("index Loss: %.6f" % res["losses"]["total_index_loss"])
