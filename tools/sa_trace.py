"""Per-SIMD timeline of the fused SA kernels from the diagnostic -DSA_STAMPS build of sa_fused.hip:
    make -C articulated-pose_amd/csrc stamps
    ANCSH_HIP_LIB=$PWD/articulated-pose_amd/csrc/build/libancsh_hip_stamps.so python tools/sa_trace.py
Every wave leaves its s_memtime phase stamps (gather | layer 1 | epilogue 1 | ... | end) and its hardware slot (HW_ID, XCC_ID) in its
output row; this script rebuilds what each SIMD did: resident waves, how often 0 / 1 / 2 waves are inside an MFMA loop, the window
from a SIMD's first wave start to its last wave end in shader clocks (the s_memtime bases differ between XCDs and shader engines, so
only same-SIMD differences are used).  profiles/r02_sa_simd_timeline.txt is its output."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import articulated_pose_amd
from articulated_pose_amd import tf_ops, _lib
from articulated_pose_amd.tf_ops.tf_sampling import farthest_point_sample_gather
from articulated_pose_amd.synthetic import make_batch
dev = "cuda:0"; B = int(os.environ.get("SA_B", "32"))
torch.manual_seed(0)
P = torch.from_numpy(make_batch(0, B, N=1024, K=3)['P']).to(dev)
_, l1 = farthest_point_sample_gather(512, P)
_, l2 = farthest_point_sample_gather(128, l1)
idx1, _ = tf_ops.query_ball_point(0.2, 64, P, l1)
idx2, _ = tf_ops.query_ball_point(0.4, 64, l1, l2)
f1 = torch.randn(B, 512, 128, device=dev)
def layers(cin, mlp):
    out = []
    for c in mlp:
        out += [torch.randn(cin, c, device=dev) / cin ** 0.5, torch.randn(c, device=dev) * .1, torch.rand(c, device=dev) + .5, torch.randn(c, device=dev) * .1]
        cin = c
    return out
def pack(W):
    W = list(W)
    for i in range(3):
        k_, n_ = W[4 * i].shape
        pk = torch.empty(_lib.lib().ancsh_sa_packed_weight_floats(k_, n_), device=dev)
        _lib.call("ancsh_sa_pack_weights", k_, n_, _lib.ptr(W[4 * i]), _lib.ptr(pk)); W[4 * i] = pk
    return W
W1 = pack(layers(3, (64, 64, 128))); W2 = pack(layers(3, (128, 128, 256)))
def sa(xyz, feats, new_xyz, idx, W, mlp):
    b, n, _ = xyz.shape; m = new_xyz.shape[1]
    ptrs = (ctypes.c_void_p * 12)(*[_lib.ptr(w) for w in W])
    out = torch.empty((b, m, mlp[2]), device=dev)
    if feats is None:
        _lib.call("ancsh_sa_module_fused", b, n, m, 64, 0, *mlp, _lib.ptr(xyz), None,
                  _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
    else:          # feats stands for the first layer's per-point partial sums
        _lib.call("ancsh_sa_module_fused_partial", b, n, m, 64, *mlp, _lib.ptr(xyz), _lib.ptr(feats),
                  _lib.ptr(new_xyz), _lib.ptr(idx), ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(out))
    return out
def timed(fn, n=400):
    for _ in range(n): o = fn()                       # loaded clock first
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): o = fn()
    e1.record(); torch.cuda.synchronize()
    return o, e0.elapsed_time(e1) * 1e3 / n           # stamps of the last launch, us per launch of the loop
NAMES = ['gather', 'L1', 'epi1', 'L2', 'epi2', 'L3', 'epi3', 'tail']
def analyse(tag, o, us, per_row):
    raw = o.reshape(-1, o.shape[-1]).view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    recs = np.concatenate([raw[:, 16 * h:16 * h + 12] for h in range(per_row)], 0)
    t0 = recs[:, 0] | (recs[:, 1] << 32)
    hw, xcc = recs[:, 2], recs[:, 3] & 0xf
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    st = recs[:, 4:12]                      # cumulative ticks since t0: gather, L1, e1, L2, e2, L3, e3, end
    dur = np.diff(np.concatenate([np.zeros((len(st), 1), np.int64), st], 1), axis=1)
    print('%s: %.1f us per launch (400 back-to-back), %d waves' % (tag, us, len(recs)))
    print('  mean phase ticks: ' + ' '.join('%s %.0f' % (n, x) for n, x in zip(NAMES, dur.mean(0))), '| lifetime %.0f' % st[:, 7].mean())
    key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
    slots = np.unique(key)
    cnt = np.bincount(np.searchsorted(slots, key))
    print('  SIMDs seen %d (xcc %d se %d sh %d cu %d), waves per SIMD min/mean/max %d/%.1f/%d' % (len(slots), len(np.unique(xcc)), len(np.unique(se)),
          len(np.unique(sh)), len(np.unique(cu)), cnt.min(), cnt.mean(), cnt.max()))
    res = {k: [] for k in ('resid', 'r0', 'r1', 'r2', 'mf0', 'mf1', 'mf2', 'busy_span')}
    for s in slots:
        m = key == s
        a = t0[m]; e = a + st[m, 7]
        lo, hi = a.min(), e.max()
        G = 2000
        grid = np.linspace(lo, hi, G, endpoint=False)
        resid = ((grid[None] >= a[:, None]) & (grid[None] < e[:, None])).sum(0)
        mf = np.zeros(G, int)
        for p0, p1 in ((0, 1), (2, 3), (4, 5)):       # MFMA loops: after stamp p0 until stamp p1
            b0 = a + st[m, p0]; b1 = a + st[m, p1]
            mf += ((grid[None] >= b0[:, None]) & (grid[None] < b1[:, None])).sum(0)
        res['resid'].append(resid.mean()); res['busy_span'].append(hi - lo)
        for k in range(3):
            res['r%d' % k].append((resid == k).mean() if k < 2 else (resid >= 2).mean())
            res['mf%d' % k].append((mf == k).mean() if k < 2 else (mf >= 2).mean())
    win = np.array(res['busy_span'])
    print('  per SIMD: window first wave start .. last wave end min/mean/max %d/%d/%d ticks = %.2f ticks/ns of the launch-to-launch time; '
          'resident waves mean %.2f; time with 0/1/2 resident: %.3f %.3f %.3f'
          % (win.min(), win.mean(), win.max(), win.mean() / us / 1e3, np.mean(res['resid']), np.mean(res['r0']), np.mean(res['r1']), np.mean(res['r2'])))
    nm = {'SA1': 392, 'SA2': 776}[tag] * cnt.mean() * 64
    print('  matrix-pipe cycles needed per SIMD (MFMAs x 64): %d = %.3f of the window' % (nm, nm / win.mean()))
    print('  time with 0/1/2+ waves inside an MFMA loop: %.3f %.3f %.3f' % (np.mean(res['mf0']), np.mean(res['mf1']), np.mean(res['mf2'])))
    # one SIMD's timeline, verbatim
    s = slots[len(slots) // 2]
    m = np.where(key == s)[0]
    m = m[np.argsort(t0[m])]
    print('  timeline of one SIMD (wave start, then cumulative stamps: gather, L1, epi1, L2, epi2, L3, epi3, end):')
    for i in m[:10]:
        print('   ', t0[i] - t0.min(), st[i].tolist())
o1, us1 = timed(lambda: sa(P, None, l1, idx1, W1, (64, 64, 128)))
analyse('SA1', o1, us1, 1)
o2, us2 = timed(lambda: sa(l1, f1, l2, idx2, W2, (128, 128, 256)))
analyse('SA2', o2, us2, 2)
