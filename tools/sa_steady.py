"""The two fused set-abstraction launches of one network (B x 1024 -> 512 -> 128, the shapes of the bench step) replayed back-to-back,
alone on the chip, long enough for the clock to settle (the power management ramps it for tens of ms under a matrix load).
SA_GROUPS networks per launch (default 2: the pipeline evaluates the ANCSH and the NPCS network of a batch in one grouped launch
per level; SA_GROUPS=1: the plain entry points).
    python tools/sa_steady.py [iterations=2000]          one line: us per launch and TF/s
    rocprofv3 --kernel-trace --stats -- python tools/sa_steady.py     -> profiles/*_kernel_stats_sa_steady.csv
Real FPS / ball-query geometry of the synthetic clouds, seeded random weights."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import articulated_pose_amd  # noqa: E402,F401
from articulated_pose_amd import _lib, tf_ops  # noqa: E402
from articulated_pose_amd.synthetic import make_batch  # noqa: E402
from articulated_pose_amd.tf_ops.tf_sampling import farthest_point_sample_gather  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev, B = "cuda:0", int(os.environ.get("SA_B", "32"))
    G = int(os.environ.get("SA_GROUPS", "2"))
    torch.manual_seed(0)
    P = torch.from_numpy(make_batch(0, B, N=1024, K=3)["P"]).to(dev)
    _, l1 = farthest_point_sample_gather(512, P)
    _, l2 = farthest_point_sample_gather(128, l1)
    idx1, _ = tf_ops.query_ball_point(0.2, 64, P, l1)
    idx2, _ = tf_ops.query_ball_point(0.4, 64, l1, l2)
    f1 = torch.randn(G * B, 512, 128, device=dev)

    def layers(cin, mlp):
        out = []
        for c in mlp:
            w = torch.randn(cin, c, device=dev) / cin ** 0.5
            pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(cin, c), device=dev)
            _lib.call("ancsh_sa_pack_weights", cin, c, _lib.ptr(w), _lib.ptr(pk))
            out += [pk, torch.randn(c, device=dev) * .1, torch.rand(c, device=dev) + .5, torch.randn(c, device=dev) * .1]
            cin = c
        return out

    def launcher(xyz, feats, new_xyz, idx, cin, mlp):
        # feats = None: a level without features.  Otherwise feats stands for the per-point partial sums of the first layer
        # (ancsh_sa_module_fused_partial; the ancsh_conv1x1 launch that produces them is not part of this loop) and cin = 3
        W = [w for _ in range(G) for w in layers(cin, mlp)]                 # one parameter set per network
        b, n, _ = xyz.shape
        m = new_xyz.shape[1]
        ptrs = (ctypes.c_void_p * (12 * G))(*[_lib.ptr(w) for w in W])
        out = torch.empty((G * b, m, mlp[2]), device=dev)
        grouped = G > 1
        if feats is None:
            name = "ancsh_sa_module_fused_grouped" if grouped else "ancsh_sa_module_fused"
            args = ((G,) if grouped else ()) + (b, n, m, 64, 0) + tuple(mlp) + (_lib.ptr(xyz), None, _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
        else:
            name = "ancsh_sa_module_fused_partial_grouped" if grouped else "ancsh_sa_module_fused_partial"
            args = ((G,) if grouped else ()) + (b, n, m, 64) + tuple(mlp) + (_lib.ptr(xyz), _lib.ptr(feats), _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
        keep = (W, ptrs, out)
        return lambda: _lib.call(name, *args), keep, 2.0 * G * b * m * 64 * (cin * mlp[0] + mlp[0] * mlp[1] + mlp[1] * mlp[2])

    res = []
    for name, (fn, keep, flops) in (("SA1 3->64->64->128", launcher(P, None, l1, idx1, 3, (64, 64, 128))),
                                    ("SA2 (3 + per-point partial sums)->128->128->256", launcher(l1, f1, l2, idx2, 3, (128, 128, 256)))):
        for _ in range(max(20, iters // 10)):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        res.append("%s: %.1f us  %.1f TF/s (%.3f of the 157.3 TF/s f32 matrix peak)" % (name, us, flops / us / 1e6, flops / us / 1e6 / 157.3))
    print(" | ".join(res) + " | %d launches each, B = %d, %d network(s) per launch" % (iters, B, G))


if __name__ == "__main__":
    main()
