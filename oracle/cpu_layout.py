"""CPU baseline in the REFERENCE'S process layout (test / benchmark infrastructure; never imported by the product).

evaluation/pose_multi_process.py:53-67 forks `os.cpu_count() - 2` worker processes, worker k solving the contiguous
slice [num_per*k, min(num_per*(k+1), n)) with num_per = int(n / workers) + 1, and joins them.  `run_layout` reproduces
exactly that layout with the CPU oracle as the per-cloud body (network forwards: oracle/ancsh_oracle.c -- the reference
has no CPU network path; pose fit: oracle/pose_oracle.py = the reference's own numpy/scipy calls with its 10000 / 200
budgets): one OS process per worker, pinned to one core, one BLAS/OpenMP thread each, wall-clock from the first spawn
to the last join.  Workers are fresh interpreters (spawned, not forked: the benchmark parent holds a HIP context).

    python -m oracle.cpu_layout --worker S E N K FULL      (internal: one worker)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve_slice(s, e, N, K, full):
    import numpy as np
    sys.path.insert(0, ROOT)
    import articulated_pose_amd  # noqa: F401   (synthetic inputs + weight tables only; no torch, no HIP)
    from articulated_pose_amd.synthetic import make_cloud, make_predictions
    from articulated_pose_amd.weights import synthetic_weights
    from oracle import net_oracle
    from oracle import pose_oracle as PO
    wa = synthetic_weights(K, seed=0)
    wn = synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1)
    t_net = t_pose = 0.0
    for cid in range(s, e):
        c = make_cloud(cid, N=N, K=K, joint_type="prismatic" if K == 4 else "revolute")
        t0 = time.time()
        net_oracle.forward(wa, c["P"][None], K)
        if full:
            net_oracle.forward(wn, c["P"][None], K, mixed_pred=False, early_split_nocs=False)
        t1 = time.time()
        if full:
            pr = make_predictions(c, K, seed=cid)
            counts = np.bincount(np.argmax(pr["instance_per_point"], 1), minlength=K)
            rs = np.random.RandomState(cid)
            sa = [PO.SampleStream([rs.randint(counts[j], size=3) for _ in range(10000)]) for j in range(K)]
            sb = [PO.SampleStream([rs.randint(counts[0 if k % 2 == 0 else j], size=3) for k in range(400)]) for j in range(1, K)]
            PO.solve_cloud(c["P"], pr["nocs_per_point"], pr["instance_per_point"], pr["joint_axis_per_point"],
                           pr["joint_cls_gt"], K, sa, sb, 0.1, 10000, 200)
        t_net += t1 - t0
        t_pose += time.time() - t1
    return t_net, t_pose


def usable_cpus():
    """CPUs this process may actually use: the scheduler affinity mask capped by the cgroup CPU quota (a container on a
    256-thread host may own 16 CPUs' worth of time: os.cpu_count() still says 256 and the reference's cpu_count-2 workers
    would then run 16x oversubscribed)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def run_layout(n_clouds, N, K, full, workers=None, timeout=600):
    """-> dict(wall_s, clouds, workers, host_cores, clouds_per_s, net_s_per_cloud, pose_s_per_cloud)."""
    host = os.cpu_count() or 1
    workers = workers or max(1, host - 2)                      # pose_multi_process.py:54
    num_per = int(n_clouds / workers) + 1                      # :55
    allowed = sorted(os.sched_getaffinity(0))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=ROOT)
    env.pop("HIP_VISIBLE_DEVICES", None)
    t0 = time.time()
    procs = []
    for k in range(workers):
        s, e = min(num_per * k, n_clouds), min(num_per * (k + 1), n_clouds)     # :61
        if e <= s:
            continue
        core = allowed[k % len(allowed)]
        procs.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_layout", "--worker", str(s), str(e), str(N), str(K),
                                       str(int(full)), str(core)], cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True))
    t_net = t_pose = 0.0
    for p in procs:
        out, _ = p.communicate(timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError("cpu_layout worker failed")
        r = json.loads(out.strip().splitlines()[-1])
        t_net += r["net_s"]
        t_pose += r["pose_s"]
    wall = time.time() - t0
    return dict(wall_s=wall, clouds=n_clouds, workers=len(procs), workers_spec=workers, host_cores=host,
                clouds_per_s=n_clouds / wall, net_s_per_cloud=t_net / n_clouds, pose_s_per_cloud=t_pose / n_clouds)


if __name__ == "__main__":
    if len(sys.argv) >= 8 and sys.argv[1] == "--worker":
        s, e, N, K, full, core = [int(x) for x in sys.argv[2:8]]
        try:
            os.sched_setaffinity(0, {core})
        except OSError:
            pass
        tn, tp = _solve_slice(s, e, N, K, bool(full))
        print(json.dumps(dict(net_s=tn, pose_s=tp)))
    else:
        print(json.dumps(run_layout(int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1024, 3, True)))
