"""Isolated timing of the backbone's mid-section (layer3, fa_layer1, fa_layer2 of both networks, 32 x 1024 clouds): the
layer-by-layer launches of rounds 3-4 against the chain launches of csrc/mid_chain.hip, each as a hipGraph replayed back to back
(loaded clock), plus per-launch HIP-event times.   python tools/mid_bench.py [B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import articulated_pose_amd  # noqa: E402,F401
from articulated_pose_amd import _lib  # noqa: E402
from articulated_pose_amd.network import Network  # noqa: E402
from articulated_pose_amd.paired import PairedNetworks  # noqa: E402
from articulated_pose_amd.tf_ops import tf_interpolate  # noqa: E402
from articulated_pose_amd.weights import synthetic_weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
K, G = 3, 2
a = Network(K, synthetic_weights(K, seed=0), "ancsh", dev)
n = Network(K, synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1), "npcs", dev)
pair = PairedNetworks([a, n])
rng = np.random.RandomState(0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
l1_xyz, l2_xyz = T(rng.uniform(-1, 1, (B, 512, 3))), T(rng.uniform(-1, 1, (B, 128, 3)))
l2_points, l1_points = T(np.abs(rng.randn(G * B, 128, 256))), T(np.abs(rng.randn(G * B, 512, 128)))
_d, fi2, fw2 = tf_interpolate.three_nn_weights(l1_xyz, l2_xyz)
L3 = [pair._layers("layer3/conv%d" % i) for i in range(3)]
F1 = [pair._layers("fa_layer1/conv_%d" % i) for i in range(2)]
F2 = [pair._layers("fa_layer2/conv_%d" % i) for i in range(2)]
args = (B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2)
flops = G * B * (128 * 2 * (259 * 256 + 256 * 512 + 512 * 1024) + 128 * 2 * (256 * 256 * 2) + 512 * 2 * (384 * 256 + 256 * 128))
st = torch.cuda.Stream()
for name, fn in (("layer by layer", pair._mid_layers), ("chains", pair._mid_chains)):
    with torch.cuda.stream(st):
        for _ in range(3):
            out = fn(*args)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            out = fn(*args)
        for _ in range(50):
            g.replay()
        st.synchronize()
        t0 = time.perf_counter()
        R = 400
        for _ in range(R):
            g.replay()
        st.synchronize()
        us = (time.perf_counter() - t0) / R * 1e6
        _lib.profile_start(lead=20)
        fn(*args)
        rec = _lib.profile_stop()
    print("%-15s %7.1f us per replay  (%.1f TFLOP/s = %.3f of 157.3 on the %0.2f GFLOP of the three levels)" % (name, us, flops / us / 1e6, flops / us / 1e6 / 157.3, flops / 1e9))
    for nm, a_, ms in rec:
        print("      %-34s %7.1f us" % (nm, ms * 1e3))
