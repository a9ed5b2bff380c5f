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


# ---- clouds + hand-built weights whose heads emit a usable segmentation and part-NOCS (coupled data flow) ----------------
def passthrough_pose_problem(K, B, N, seed=0, nocs_scale=6.0):
    """Clouds + weights for which the NETWORKS THEMSELVES emit a usable segmentation and part-NOCS, so the pose stage can
    be fed by them (AncshPipeline(couple=True)) and still has a known answer.

    fa_layer3 receives the raw xyz as skip features (pointnet_plusplus/architectures.py:84-86).  The hand-built weights
    route relu(+-x), relu(+-y), relu(+-z) through fa_layer3/conv_0..2 and fc1 (identity on 6 channels, zero elsewhere,
    BN scale 1) so that the trunk feature carries the point's coordinates; every layer before that keeps seeded random
    weights (their outputs are multiplied by zero columns).  Heads of the NPCS network:
        fc2_0 (part logits)  = g * (plane_j . P + d_j)          -> parts = slabs along x
        fc2_1 (NOCS logits)  = 4 * (R_j^T (P - t_j) / s_j - 0.5) -> sigmoid(z) = 0.5 + z/4 - z^3/48 ~ the exact part-NOCS
    (|z| <= ~0.25, so the cubic term is < 4e-4 of a NOCS unit), and of the ANCSH network: fc4_0 bias = atanh(joint axis).
    Returns P (B,N,3), cls (B,N), the articulated pose R (K,3,3) / s (K) / t (K,3) shared by all clouds, joint_axis, and
    the two weight dicts.
    """
    from .weights import BN_EPS, synthetic_weights
    rng = np.random.RandomState(seed)
    q = rng.randn(4); q /= np.linalg.norm(q); w, x, y, z = q
    R0 = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    u = rng.randn(3); u /= np.linalg.norm(u)
    # parts = slabs along x in [-0.45, 0.45], all inside the unit cube
    edges = np.linspace(-0.45, 0.45, K + 1)
    centres = np.stack([np.array([0.5 * (edges[j] + edges[j + 1]), 0.0, 0.0]) for j in range(K)])
    Rs = [R0] + [R0 @ _axis_rot(u, rng.uniform(-1.0, 1.0)) for _ in range(1, K)]
    ss = rng.uniform(0.9, 1.1, K) * nocs_scale
    ts = [centres[j] - ss[j] * Rs[j] @ np.full(3, 0.5) for j in range(K)]         # NOCS (0.5,0.5,0.5) -> part centre
    P = np.zeros((B, N, 3), np.float32)
    cls = np.zeros((B, N), np.int64)
    for b in range(B):
        lab = rng.randint(0, K, N)
        lab[:K * 40] = np.repeat(np.arange(K), 40)                                   # every part populated
        half = (edges[1] - edges[0]) * 0.5 * 0.9                                      # stay clear of the slab boundaries
        pts = centres[lab] + np.stack([rng.uniform(-half, half, N), rng.uniform(-0.25, 0.25, N), rng.uniform(-0.25, 0.25, N)], 1)
        P[b], cls[b] = pts.astype(np.float32), lab

    def build(mixed):
        wts = synthetic_weights(K, mixed_pred=mixed, early_split_nocs=mixed, seed=3 if mixed else 4)
        e = "SPFN/est_net/"
        one = np.float32(1.0)

        def identity_bn(scope, cout):
            var = np.full(cout, one - np.float32(BN_EPS), np.float32)
            wts[scope + "/bn/moving_variance"] = var
            wts[scope + "/bn/gamma"] = np.sqrt(var + np.float32(BN_EPS)).astype(np.float32)
            wts[scope + "/bn/beta"] = np.zeros(cout, np.float32)
            wts[scope + "/bn/moving_mean"] = np.zeros(cout, np.float32)
            wts[scope + "/biases"] = np.zeros(cout, np.float32)

        k0 = np.zeros((1, 1, 131, 128), np.float32)
        for i in range(3):
            k0[0, 0, 128 + i, 2 * i] = 1.0
            k0[0, 0, 128 + i, 2 * i + 1] = -1.0
        wts[e + "fa_layer3/conv_0/weights"] = k0
        identity_bn(e + "fa_layer3/conv_0", 128)
        eye6 = np.zeros((128, 128), np.float32)
        eye6[:6, :6] = np.eye(6)
        for sc, shape in ((e + "fa_layer3/conv_1", (1, 1, 128, 128)), (e + "fa_layer3/conv_2", (1, 1, 128, 128)), (e + "fc1", (1, 128, 128))):
            wts[sc + "/weights"] = eye6.reshape(shape).copy()
            identity_bn(sc, 128)

        def linear_head(scope, A, bias):
            """logits = A @ P + bias through the (relu(+c), relu(-c)) channels."""
            cout = A.shape[0]
            kk = np.zeros((1, 128, cout), np.float32)
            for i in range(3):
                kk[0, 2 * i, :] = A[:, i]
                kk[0, 2 * i + 1, :] = -A[:, i]
            wts[scope + "/weights"] = kk
            wts[scope + "/biases"] = np.asarray(bias, np.float32)

        n = "SPFN/nocs_net/"
        # slab logits: -g * (x - centre_j)^2 up to a common term = g * (2 c_j x - c_j^2)
        g = 200.0
        A = np.zeros((K, 3)); A[:, 0] = g * 2 * centres[:, 0]
        seg_head = (A, -g * centres[:, 0] ** 2)
        M = np.concatenate([4.0 / ss[j] * Rs[j].T for j in range(K)])                  # (3K, 3)
        bias = np.concatenate([-4.0 / ss[j] * Rs[j].T @ ts[j] - 2.0 for j in range(K)])
        if mixed:      # ANCSH: fc2_1 sits behind fc11_1 (no BN, no activation): identity on the six channels
            linear_head(n + "fc2_0", *seg_head)
            wts[n + "fc11_1/weights"] = eye6.reshape(1, 128, 128).copy()
            wts[n + "fc11_1/biases"] = np.zeros(128, np.float32)
            linear_head(n + "fc2_1", M, bias)
            j = "SPFN/joint_net/"
            wts[j + "fc4_0/weights"] = np.zeros((1, 128, 3), np.float32)
            wts[j + "fc4_0/biases"] = np.arctanh(u).astype(np.float32)
        else:
            linear_head(n + "fc2_0", *seg_head)
            linear_head(n + "fc2_1", M, bias)
        return wts

    return dict(P=P, cls=cls, joint_axis=u, R=np.stack(Rs), s=ss, t=np.stack(ts), w_ancsh=build(True), w_npcs=build(False))
