"""Work distribution of the stage-B LM fits of one batch (32 clouds x 2 joints x 200 hypotheses), from a DIAGNOSTIC build of the
library (make -C articulated-pose_amd/csrc EXTRA=-DLM_COUNT; rebuild without it afterwards): per fit the MINPACK evaluation
count, the trips of the lmdif loop body and the Cholesky factorisations inside lmpar.  -> profiles/*_lm_fit_lengths.txt"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import articulated_pose_amd
from articulated_pose_amd.pose import PoseSolver
from articulated_pose_amd.synthetic import make_cloud, make_predictions
K, N, B = 3, 1024, 32
clouds = [make_cloud(i, N=N, K=K) for i in range(B)]
preds = [make_predictions(c, K, seed=i) for i, c in enumerate(clouds)]
dev = 'cuda:0'
solver = PoseSolver(K, 0.1, 10000, 200, dev, want_lm_stat=True, lm_schedule="throughput")
args = [torch.from_numpy(np.stack(x)).to(dev) for x in ([c["P"] for c in clouds], [p["nocs_per_point"] for p in preds], [p["instance_per_point"] for p in preds], [p["joint_axis_per_point"] for p in preds], [p["joint_cls_gt"].astype(np.int32) for p in preds])]
sol = solver.solve(*args, seed=1)
s = sol["lm_stat"].cpu().numpy().reshape(-1, 2)
info, trips, nchol, nfev = s[:, 0] & 15, (s[:, 0] >> 4) & 8191, (s[:, 0] >> 17) & 16383, s[:, 1]
print("fits", len(nfev), "nfev mean %.1f  trips mean %.1f  chol mean %.1f  chol/trip %.2f" % (nfev.mean(), trips.mean(), nchol.mean(), nchol.sum() / max(1, trips.sum())))
long = nfev > 1000
print("long fits (nfev > 1000):", long.sum(), " nfev mean %.0f trips mean %.0f chol mean %.0f chol/trip %.2f accepted/trip %.2f" % (
    nfev[long].mean(), trips[long].mean(), nchol[long].mean(), nchol[long].sum() / trips[long].sum(), ((nfev[long] - 1 - trips[long]) / 6 / trips[long]).mean()))
print("share of all Cholesky factorisations in long fits: %.2f ; share of nfev: %.2f" % (nchol[long].sum() / nchol.sum(), nfev[long].sum() / nfev.sum()))
w = nfev.reshape(-1, 200)
print("problems with a long fit: %d of %d; long fits per problem mean %.2f" % ((w > 1000).any(1).sum(), len(w), (w > 1000).sum(1).mean()))
print("info hist", np.bincount(info))
for o in np.argsort(nfev)[::-1][:8]:
    print("  fit", o, "nfev", nfev[o], "trips", trips[o], "chol", nchol[o], "info", info[o])
