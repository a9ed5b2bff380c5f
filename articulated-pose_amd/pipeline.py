"""End-to-end hot path for a batch of depth point clouds on one MI355X:

    ANCSH forward  (joint axes / association heads)        main.py --test --nocs_type=ancsh
    NPCS  forward  (part-NOCS + part masks, the "baseline" network the pose stage reads when
                    USE_BASELINE, evaluation/parallel_ancsh_pose.py:232-237)   main.py --test --nocs_type=npcs
    pose fit       (per-part RANSAC + articulated LM)      evaluation/pose_multi_process.py

The whole step is ~100 asynchronous launches on one HIP stream; `AncshPipeline` captures them once
into a hipGraph and replays it per batch.
"""
import numpy as np
import torch

from .network import Network
from .pose import PoseSolver


class AncshPipeline(object):
    """step() consumes the resident input buffers and returns the pose records of the batch.

    couple=True : the pose stage reads the networks' own outputs (production data flow).
    couple=False: the pose stage reads `pred_*` buffers supplied by the caller -- used by the benchmark,
                  where random-init networks (no checkpoint ships with the reference) would hand the
                  fitter degenerate parts; every stage still runs inside the step."""

    def __init__(self, num_parts, weights_ancsh, weights_npcs, batch_size, num_points, device="cuda:0",
                 inlier_th=0.1, niter_a=10000, niter_b=200, couple=True, use_graph=True, seed=0):
        self.K, self.B, self.N = num_parts, batch_size, num_points
        self.device = torch.device(device)
        self.ancsh = Network(num_parts, weights_ancsh, "ancsh", device)
        self.npcs = Network(num_parts, weights_npcs, "npcs", device)
        self.solver = PoseSolver(num_parts, inlier_th, niter_a, niter_b, device)
        self.couple, self.seed = couple, seed
        B, N, K = batch_size, num_points, num_parts
        f = dict(dtype=torch.float32, device=self.device)
        self.P = torch.zeros((B, N, 3), **f)
        self.joint_cls = torch.zeros((B, N), dtype=torch.int32, device=self.device)
        self.pred_nocs = torch.zeros((B, N, 3 * K), **f)
        self.pred_mask = torch.zeros((B, N, K), **f)
        self.pred_axis = torch.zeros((B, N, 3), **f)
        self.stream = torch.cuda.Stream(device=self.device)
        self.graph = None
        self.out = None
        self._use_graph = use_graph

    def load_inputs(self, P, joint_cls, pred=None):
        self.P.copy_(torch.as_tensor(P))
        self.joint_cls.copy_(torch.as_tensor(np.asarray(joint_cls, np.int32)) if not torch.is_tensor(joint_cls) else joint_cls)
        if pred is not None:
            self.pred_nocs.copy_(torch.as_tensor(pred["nocs_per_point"]))
            self.pred_mask.copy_(torch.as_tensor(pred["instance_per_point"]))
            self.pred_axis.copy_(torch.as_tensor(pred["joint_axis_per_point"]))

    def _run(self):
        a = self.ancsh.predict(self.P)
        n = self.npcs.predict(self.P)
        if self.couple:
            nocs, mask, axis = n["nocs_per_point"], n["W"], a["joint_axis_per_point"]
        else:
            nocs, mask, axis = self.pred_nocs, self.pred_mask, self.pred_axis
        sol = self.solver.solve(self.P, nocs, mask, axis, self.joint_cls, seed=self.seed)
        record = torch.cat([sol["baseline"], sol["nonlinear"]], dim=2)      # (B, K, 26) float64
        return dict(ancsh=a, npcs=n, pose=sol, record=record)

    def prepare(self):
        with torch.cuda.stream(self.stream):
            for _ in range(2):
                self.out = self._run()
        self.stream.synchronize()
        if self._use_graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.out = self._run()
        return self

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.out = self._run()
        return self.out
