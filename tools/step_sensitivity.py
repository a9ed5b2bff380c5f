"""What the step is sensitive to (round 4): the pipeline of bench.py with parts of the step removed or their budgets cut.
    python tools/step_sensitivity.py full|net|pose NITER_A NITER_B      three pipelines in one process (the FIRST is the clean figure)
    python tools/step_sensitivity.py inst SLOTS                         four full pipelines one after the other in one process
Results: profiles/r04_step_sensitivity.txt."""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")     # one hardware queue per batch in flight: before HIP initialises
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import articulated_pose_amd  # noqa
from articulated_pose_amd.pipeline import AncshPipeline
from articulated_pose_amd.synthetic import make_cloud, make_predictions
from articulated_pose_amd.weights import synthetic_weights
K, B, N = 3, 32, 1024
dev = "cuda:0"
wa, wn = synthetic_weights(K, seed=0), synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1)
clouds = [make_cloud(i, N=N, K=K) for i in range(B)]
P = np.stack([c["P"] for c in clouds]).astype(np.float32)
preds = [make_predictions(c, K, seed=i) for i, c in enumerate(clouds)]

class Pipe(AncshPipeline):
    mode = "full"
    def _run(self, sl=None):
        sl = sl or self.slots[0]
        from articulated_pose_amd.pointnet_util import Geometry
        out = {}
        if self.mode in ("full", "net"):
            geom = Geometry()
            a, n = self.paired.predict(sl.P, geom)
            out.update(ancsh=a, npcs=n)
        if self.mode in ("full", "pose"):
            sol = self.solver.solve(sl.P, sl.pred_nocs, sl.pred_mask, sl.pred_axis, sl.joint_cls, draws_a=None, draws_b=None, seed=self.seed)
            out["record"] = torch.cat([sol["baseline"], sol["nonlinear"]], dim=2)
        return out

def run(mode, na, nb, slots=20, steps=384):
    Pipe.mode = mode
    pipe = Pipe(K, wa, wn, B, N, dev, couple=False, use_graph=True, seed=0, slots=slots, niter_a=na, niter_b=nb)
    pipe.load_inputs(P, np.stack([p["joint_cls_gt"] for p in preds]),
                     {k: np.stack([p[k] for p in preds]) for k in ("nocs_per_point", "instance_per_point", "joint_axis_per_point")})
    pipe.prepare()
    for _ in range(64): pipe.step()
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): pipe.step()
    pipe.synchronize(); torch.cuda.synchronize()
    print("%-5s niter_a %6d niter_b %4d slots %2d  %.4f ms/step" % (mode, na, nb, slots, (time.perf_counter() - t0) / steps * 1e3), flush=True)

import gc
if sys.argv[1] == "inst":
    slots = int(sys.argv[2])
    for rep in range(4):
        run("full", 10000, 200, slots=slots, steps=512)
        gc.collect(); torch.cuda.empty_cache()
    sys.exit(0)
mode, na, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for rep in range(3):
    run(mode, na, nb, steps=512)
