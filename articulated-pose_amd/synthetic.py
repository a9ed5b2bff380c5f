"""Synthetic articulated point clouds and synthetic network predictions: there is no
network access for the shape2motion/SAPIEN data or the pretrained checkpoints, so parity tests and
benchmarks use seeded synthetic inputs of the reference's shapes.

A cloud = K rigid parts; part j = n_j points on the surface of an axis-aligned box inside its NOCS
cube; camera-space points P = s_j * R_j * nocs_j + t_j with R_j = R_0 * Rot(u, theta_j) (revolute,
kinematically consistent: R_0 u = R_j u) or R_j = R_0, t_j = t_0 + d_j * R_0 u (prismatic); the
whole cloud is rescaled into the unit cube like the reference loader (pts * norm_factor,
lib/dataset.py:351).  seed = 1234 + cloud_id mirrors main.py:20.
"""
import numpy as np


def _rand_rot(rng):
    q = rng.randn(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _axis_rot(u, th):
    ux = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(th) * ux + (1 - np.cos(th)) * (ux @ ux)


def _box_surface(rng, n, ext):
    """n points uniform on the surface of the box [0.5 - ext/2, 0.5 + ext/2]."""
    p = rng.uniform(-0.5, 0.5, (n, 3))
    face = rng.randint(0, 3, n)
    sign = rng.randint(0, 2, n) - 0.5
    p[np.arange(n), face] = sign
    return 0.5 + p * ext


def make_cloud(cloud_id, N=1024, K=3, joint_type='revolute'):
    rng = np.random.RandomState(1234 + cloud_id)
    frac = rng.dirichlet([2.0] * K)
    n = np.maximum(64, np.round(frac * N).astype(int))
    while n.sum() != N:
        n[np.argmax(n)] += np.sign(N - n.sum())
    R0 = _rand_rot(rng)
    u = rng.randn(3)
    u /= np.linalg.norm(u)
    s = rng.uniform(0.6, 1.2, K)
    t0 = rng.uniform(-0.2, 0.2, 3)
    P, nocs, cls, Rs, ts = [], [], [], [], []
    for j in range(K):
        ext = rng.uniform(0.3, 1.0, 3)
        x = _box_surface(rng, n[j], ext)
        if joint_type == 'revolute':
            Rj = R0 if j == 0 else R0 @ _axis_rot(u, rng.uniform(-1.2, 1.2))
            tj = t0 + rng.uniform(-0.3, 0.3, 3) * (j > 0)
        else:
            Rj = R0
            tj = t0 + (rng.uniform(-0.6, 0.6) * (R0 @ u) + rng.uniform(-0.3, 0.3, 3)) * (j > 0)
        P.append(s[j] * x @ Rj.T + tj)
        nocs.append(x)
        cls.append(np.full(n[j], j))
        Rs.append(Rj)
        ts.append(tj)
    P = np.concatenate(P)
    ctr = 0.5 * (P.max(0) + P.min(0))
    g = 1.0 / np.linalg.norm(P.max(0) - P.min(0))          # unit-diagonal normalisation
    P = (P - ctr) * g
    perm = rng.permutation(N)
    out = dict(P=P[perm].astype(np.float32), nocs_gt=np.concatenate(nocs)[perm].astype(np.float32),
               cls_gt=np.concatenate(cls)[perm].astype(np.int64), joint_axis=u.astype(np.float32),
               R=np.stack(Rs), s=s * g, t=np.stack([(ts[j] - ctr) * g for j in range(K)]), n_parts=n)
    return out


def make_batch(first_id, B, N=1024, K=3, joint_type='revolute'):
    cl = [make_cloud(first_id + i, N, K, joint_type) for i in range(B)]
    return {k: np.stack([c[k] for c in cl]) for k in cl[0]}


def make_predictions(cloud, K, seed=0, noise=0.01, outlier=0.10, flip=0.05, axis_noise=0.02):
    """Synthetic stand-ins for the network outputs the pose stage consumes:
    nocs (N,3K) = GT part-NOCS (+N(0,noise), `outlier` fraction uniform), W (N,K) one-hot GT with
    `flip` label flips, joint_axis_per_point (N,3) = u + N(0,axis_noise), joint_cls_gt (N,)."""
    rng = np.random.RandomState(seed)
    N = cloud['P'].shape[0]
    cls = cloud['cls_gt'].copy()
    nocs_part = cloud['nocs_gt'] + rng.randn(N, 3) * noise
    out = rng.rand(N) < outlier
    nocs_part[out] = rng.uniform(0, 1, (int(out.sum()), 3))
    nocs = rng.uniform(0, 1, (N, 3 * K))
    for j in range(K):
        nocs[:, 3 * j:3 * j + 3] = nocs_part       # every slot carries the point's own-part prediction
    lab = cls.copy()
    fl = rng.rand(N) < flip
    lab[fl] = rng.randint(0, K, int(fl.sum()))
    W = np.full((N, K), 0.05 / max(K - 1, 1))
    W[np.arange(N), lab] = 0.95
    axis = cloud['joint_axis'][None] + rng.randn(N, 3) * axis_noise
    joint_cls = cls.copy()                            # points associated with joint j = points of part j
    return dict(nocs_per_point=nocs.astype(np.float32), instance_per_point=W.astype(np.float32),
                joint_axis_per_point=axis.astype(np.float32), joint_cls_gt=joint_cls.astype(np.int64))
