#!/usr/bin/env python
"""Where the ball query's time goes (round 4).  Two measurements on the set-abstraction level-1 query (r = 0.2, 64 samples) at
16 x 2048 and 32 x 1024, plus a sparse case (r = 0.05 / 0.08: every query scans the whole cloud):

  python tools/bq_phase.py                                   per-launch time of each operator of the op-level graph, every schedule
                                                             (each in its own process: the schedule is read once), in-cache graph replay
  make -C articulated-pose_amd/csrc bqstamps
  ANCSH_HIP_LIB=$PWD/articulated-pose_amd/csrc/build/libancsh_hip_bqstamps.so python tools/bq_phase.py --stamps
                                                             s_memtime phase stamps + steps executed per wave of the wave-per-two-queries
                                                             kernel (diagnostic -DBQL_STAMPS build; shader clocks, ~2.4 per ns)
profiles/r04_ball_query_phases.txt is the output of both."""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import articulated_pose_amd  # noqa: E402,F401
from articulated_pose_amd import _lib, tf_ops  # noqa: E402
from articulated_pose_amd.synthetic import make_batch  # noqa: E402
from articulated_pose_amd.tf_ops.tf_sampling import farthest_point_sample_gather  # noqa: E402


def gtime(fn, reps=20, inner=20):
    """us per launch: `inner` launches captured in a hipGraph, replayed `reps` times between two events on the graph's stream"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); fn()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        keep = [fn() for _ in range(inner)]
    with torch.cuda.stream(st):
        for _ in range(10):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
    st.synchronize()
    del keep
    return round(e0.elapsed_time(e1) / (reps * inner) * 1e3, 2)


def one_schedule():
    dev = torch.device("cuda:0")
    sched = os.environ.get("ANCSH_BQ_SCHEDULE", "wave (default)")
    for B, N, r1 in ((16, 2048, 0.2), (32, 1024, 0.2), (16, 2048, 0.05), (32, 1024, 0.08)):
        P = torch.from_numpy(make_batch(0, B, N=N, K=3)["P"]).to(dev)
        _, l1 = farthest_point_sample_gather(512, P)
        _, l2 = farthest_point_sample_gather(128, l1)
        idx1, c1 = tf_ops.query_ball_point(r1, 64, P, l1)
        pop = float((torch.cdist(l1[:2], P[:2]) < r1).float().sum(2).mean())
        row = {"bq1": gtime(lambda: tf_ops.query_ball_point(r1, 64, P, l1)), "bq2": gtime(lambda: tf_ops.query_ball_point(2 * r1, 64, l1, l2)),
               "bq_multi": gtime(lambda: tf_ops.query_ball_point_multi([(r1, 64, P, l1), (2 * r1, 64, l1, l2)])),
               "bq_group_xyz_multi": gtime(lambda: tf_ops.query_ball_group_xyz_multi([(r1, 64, P, l1), (2 * r1, 64, l1, l2)]))}
        if r1 == 0.2 and not sched.startswith("lanes"):
            idx2, _ = tf_ops.query_ball_point(0.4, 64, l1, l2)
            f1 = torch.randn(B, 512, 128, device=dev)
            row.update(group_xyz1=gtime(lambda: tf_ops.group_point(P, idx1)), group_xyz2=gtime(lambda: tf_ops.group_point(l1, idx2)),
                       group_feat=gtime(lambda: tf_ops.group_point(f1, idx2), inner=8),
                       group_multi_all3=gtime(lambda: tf_ops.group_point_multi([(P, idx1), (l1, idx2), (f1, idx2)]), inner=8))
        print("%-14s %2d x %4d r=%.2f  mean ball population %5.0f  mean pts_cnt %4.1f  us per launch: %s"
              % (sched, B, N, r1, pop, float(c1.float().mean()), row), flush=True)


def stamps():
    dev = torch.device("cuda:0")
    for B, N in ((16, 2048), (32, 1024)):
        P = torch.from_numpy(make_batch(0, B, N=N, K=3)["P"]).to(dev)
        _, l1 = farthest_point_sample_gather(512, P)
        for _ in range(3):
            tf_ops.query_ball_point(0.2, 64, P, l1)
        torch.cuda.synchronize()
        buf = np.zeros(4096 * 4 * 8, np.uint64)
        assert _lib.lib().ancsh_debug_bqw_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        s = buf.reshape(4096, 4, 8)[:min(4096, B * 512 // 8)].astype(np.int64)
        d = {"stage cloud into LDS": s[..., 1] - s[..., 0], "query + first candidates": s[..., 2] - s[..., 1], "scan": s[..., 3] - s[..., 2],
             "fill + counts": s[..., 4] - s[..., 3], "whole wave": s[..., 4] - s[..., 0]}
        print("wave-per-two-queries kernel, %d x %d, r = 0.2, 64 samples: shader clocks per wave (median / max over %d waves)" % (B, N, s.shape[0] * 4))
        for k, v in d.items():
            print("   %-26s %6d / %6d" % (k, int(np.median(v)), int(v.max())))
        steps = s[..., 5].ravel().astype(int)
        print("   128-candidate steps per wave: median %d, max %d; histogram (waves with 0..%d steps): %s"
              % (int(np.median(steps)), steps.max(), N // 128, np.bincount(steps, minlength=N // 128 + 1).tolist()), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--one", action="store_true", help="(internal) this process measures the schedule in ANCSH_BQ_SCHEDULE")
    a = ap.parse_args()
    if a.stamps:
        return stamps()
    if a.one:
        return one_schedule()
    for sched in ("", "lanes0", "lanes1", "lanes2"):
        env = dict(os.environ)
        env.pop("ANCSH_BQ_SCHEDULE", None)
        if sched:
            env["ANCSH_BQ_SCHEDULE"] = sched
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, check=False)


if __name__ == "__main__":
    main()
