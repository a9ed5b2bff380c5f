"""Host side of the test-time losses (no GPU): which terms enter total_loss and which fields test_loss.txt gets depend on the
flags main.py sets per --nocs_type (main.py:31-34,42-52): 'ancsh' switches pred_joint / pred_joint_ind / early_split on, 'npcs'
leaves the argparse defaults (False), so the NPCS total is 10*nocs + miou and its line is 'Total Loss, MIoU Loss, nocs Loss'
(lib/network.py:162-169, :228-243)."""
import numpy as np
import torch

from articulated_pose_amd import loss as L
from articulated_pose_amd.network import Network
from oracle import loss_oracle as LO


def _loss_dict(seed, B=5, K=3, mixed=False):
    r = np.random.RandomState(seed)
    ld = {"nocs_loss": r.rand(B), "miou_loss": r.rand(B, K), "heatmap_loss": r.rand(B), "unitvec_loss": r.rand(B),
          "orient_loss": r.rand(B), "index_loss": r.rand(B, 3)}
    if mixed:
        ld["gocs_loss"] = r.rand(B)
    return {k: v.astype(np.float32) for k, v in ld.items()}


def test_network_flags_follow_main_py():
    a, n = Network(3, {}, "ancsh", "cpu"), Network(3, {}, "npcs", "cpu")
    assert (a.is_mixed, a.pred_joint, a.pred_joint_ind, a.early_split, a.early_split_nocs) == (True,) * 5
    assert (n.is_mixed, n.pred_joint, n.pred_joint_ind, n.early_split, n.early_split_nocs) == (False,) * 5
    assert Network(3, {}, "npcs", "cpu", pred_joint=True).pred_joint           # --pred_joint on the command line


def test_collect_losses_and_line_per_configuration():
    for mixed, pj, pji, es in ((True, True, True, True), (False, False, False, False), (False, True, False, False), (False, True, True, True)):
        ld = _loss_dict(int(mixed) + 2 * int(pj) + 4 * int(pji), mixed=mixed)
        got = L.collect_losses({k: torch.from_numpy(v) for k, v in ld.items()}, mixed, pj, pji)
        want = LO.collect_losses(ld, mixed, pj, pji)
        for k in want:
            assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (k, got[k], want[k])
        keys = L.reported_keys(mixed, pj, es, pji)
        line = L.format_loss_result(got, mixed, pj, es, pji)
        assert [f.split(":")[0] for f in line.split(", ")] == [
            {"total_loss": "Total Loss", "total_miou_loss": "MIoU Loss", "total_nocs_loss": "nocs Loss", "total_gocs_loss": "gocs Loss",
             "total_heatmap_loss": "heatmap Loss", "total_unitvec_loss": "unitvec Loss", "total_orient_loss": "orient Loss",
             "total_index_loss": "index Loss"}[k] for k in keys]
    # the NPCS baseline of main.py --nocs_type=npcs: three fields, total = 10 * nocs + miou
    ld = _loss_dict(9)
    got = L.collect_losses({k: torch.from_numpy(v) for k, v in ld.items()}, False, False, False)
    assert abs(got["total_loss"] - (10.0 * float(ld["nocs_loss"].astype(np.float64).mean()) + float(ld["miou_loss"].astype(np.float64).mean()))) < 1e-6
    line = L.format_loss_result(got, False, False, False, False)
    assert line == "Total Loss: {:6f}, MIoU Loss: {:6f}, nocs Loss: {:6f}".format(got["total_loss"], got["total_miou_loss"], got["total_nocs_loss"])
    assert L.reported_keys(False, False, False, False) == ["total_loss", "total_miou_loss", "total_nocs_loss"]
