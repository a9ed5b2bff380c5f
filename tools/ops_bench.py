#!/usr/bin/env python
"""The op-level ball_query + group figure of bench.py on its own (same function, same graph), for profiling:
    python tools/ops_bench.py [--mode five|multi|fused] [--sets 1|12] [--batch 32] [--npoints 1024]
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/ops_bench.py --sets 12       (tools/capture_profiles.sh ops)
--sets 1 replays one ~180 MB operand set (stays inside the 256 MiB Infinity Cache); --sets 12 walks twelve independent
sets per replay, i.e. every byte comes from HBM."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from articulated_pose_amd.synthetic import make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="five", choices=["five", "multi", "fused", "fused_multi"])
    ap.add_argument("--sets", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--npoints", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    P = torch.from_numpy(make_batch(0, a.batch, N=a.npoints, K=3)["P"]).to(dev)
    r = bench.op_level_ball_group(P, a.batch, a.npoints, dev, a.mode, sets=a.sets, reps=a.reps)
    print(json.dumps(r))


if __name__ == "__main__":
    main()
