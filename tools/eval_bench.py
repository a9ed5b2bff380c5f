"""Timing of the evaluation drop-ins' kernels at the size of a real test split (3480 frames of 1024 points, K = 3: global_info.py,
eyeglasses test_size): ancsh_part_extents (HBM-bound: mask + NOCS + P read once per part) and ancsh_iou_3d (50^3 grid per box pair).
    python tools/eval_bench.py            -> profiles/r04_eval_scripts.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import articulated_pose_amd  # noqa
from articulated_pose_amd.pose import evaluation as E, metrics as M

dev = "cuda:0"
F, N, K = 3480, 1024, 3
g = torch.Generator(device="cpu").manual_seed(0)
nocs = torch.rand(F, N, 3 * K, generator=g).to(dev)
mask = torch.rand(F, N, K, generator=g).to(dev)
P = torch.randn(F, N, 3, generator=g).to(dev)
q, _ = np.linalg.qr(np.random.RandomState(0).randn(F, 3, 3))
t0 = np.random.RandomState(1).randn(F, 3)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


qd, td = torch.as_tensor(q, device=dev), torch.as_tensor(t0, device=dev)
us = timed(lambda: E.part_extents(nocs, mask, P, qd, td))
alg = F * N * (K * 4 + 3 * K * 4 + 12)           # every byte of mask, NOCS and P once
print("ancsh_part_extents  %d frames x %d points, K = %d: %8.1f us per call algorithmic %.1f MB -> %.0f GB/s = %.2f of 8 TB/s"
      % (F, N, K, us, alg / 1e6, alg / us / 1e3, alg / us / 1e3 / 8000))
sc = torch.rand(F * K, 3, dtype=torch.float64, device=dev) * 0.5 + 0.3
s = torch.rand(F * K, dtype=torch.float64, device=dev) * 0.4 + 0.8
R = torch.as_tensor(np.repeat(q, K, axis=0), device=dev)
t = torch.as_tensor(np.repeat(t0, K, axis=0), device=dev) * 0.05
b1 = M.amodal_boxes(sc, s, R, t)
b2 = M.amodal_boxes(sc * 1.05, s, R, t + 0.01)
us = timed(lambda: M.iou_3d_batch(b1, b2), reps=5)
print("ancsh_iou_3d        %d box pairs, 50^3 grid points each in two boxes: %8.1f us per call -> %.2f G point-in-box tests/s"
      % (F * K, us, F * K * 125000 * 2 / us / 1e3))
