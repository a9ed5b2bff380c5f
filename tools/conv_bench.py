import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import articulated_pose_amd
from articulated_pose_amd import _lib
dev = 'cuda:0'
def t(fn, n=200):
    for _ in range(200): fn()          # loaded clock
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, cin, cout, pool in ((4096, 259, 256, 0), (4096, 256, 256, 0), (4096, 256, 512, 0), (4096, 512, 1024, 128), (16384, 128, 128, 0), (16384, 384, 256, 0), (16384, 256, 128, 0), (65536, 256, 256, 0), (262144, 256, 256, 0), (65536, 2048, 256, 0), (65536, 64, 256, 0)):
    ld = (cin + 3) // 4 * 4
    x = torch.randn(rows, ld, device=dev); W = torch.randn(cin, cout, device=dev) / 16
    b, sc, sh = [torch.randn(cout, device=dev) for _ in range(3)]
    pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(cin, cout), device=dev)
    _lib.call("ancsh_sa_pack_weights", cin, cout, _lib.ptr(W), _lib.ptr(pk))
    y = torch.empty(rows // pool if pool else rows, cout, device=dev)
    t_old = t(lambda: _lib.call("ancsh_conv1x1", rows, cin, cout, _lib.ptr(x), ld, _lib.ptr(W), _lib.ptr(b), _lib.ptr(sc), _lib.ptr(sh), 1, _lib.ptr(y), cout, pool))
    t_new = t(lambda: _lib.call("ancsh_conv1x1_packed", rows, cin, cout, _lib.ptr(x), ld, _lib.ptr(pk), _lib.ptr(b), _lib.ptr(sc), _lib.ptr(sh), 1, _lib.ptr(y), cout, pool, None, 0))
    fl = 2.0 * rows * cin * cout
    print('%7d x %4d -> %4d pool %3d : tiled %7.1f us %6.1f TF/s | packed %7.1f us %6.1f TF/s' % (rows, cin, cout, pool, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6))
