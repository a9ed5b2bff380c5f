"""stage-A scoring kernel duration against the part size: separates the per-hypothesis prologue (3-point model) from the per-point cost"""
import sys, os
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/articulated-pose_amd") else os.getcwd())
import numpy as np, torch
import articulated_pose_amd  # noqa
from articulated_pose_amd.pose import ransac_single_batch
dev = "cuda:0"
rng = np.random.RandomState(0)
for n in (8, 64, 128, 341, 682, 1024):
    nprob, niter = 96, 10000
    off = (np.arange(nprob + 1) * n).astype(np.int32)
    src = (rng.rand(nprob * n, 3).astype(np.float32) - 0.5)
    tgt = (1.1 * src + 0.1 + rng.randn(nprob * n, 3).astype(np.float32) * 0.02).astype(np.float32)
    o, s, t = (torch.from_numpy(a).to(dev) for a in (off, src, tgt))
    for _ in range(3):
        ransac_single_batch(o, s, t, 0.1, niter, None, seed=1, max_n=n)
    torch.cuda.synchronize()
print("done")
