"""Weights of the ANCSH / NPCS networks as a flat dict keyed by the reference's TF variable names.

Variable names (scope 'SPFN', lib/network.py:58):
  SPFN/est_net/layer{1,2,3}/conv{0,1,2}/{weights,biases,bn/{beta,gamma,moving_mean,moving_variance}}
      (pointnet_util.py:114,128 -> tf_util.conv2d :155-184; kernel [1,1,cin,cout])
  SPFN/est_net/fa_layer{1,2,3}/conv_{i}/...           (pointnet_util.py:228-234)
  SPFN/est_net/fc1/...                                (architectures.py:89; conv1d kernel [1,cin,cout])
  SPFN/nocs_net/{fc2_0..,fc11_1}/{weights,biases}     (lib/architecture.py:105-120; no BN, no activation)
  SPFN/joint_net/{fc3_0,fc3_1}/... (+bn), fc4_{0..3}  (lib/architecture.py:195-206)
No pretrained checkpoint ships with the reference (README.md:80-92 links only), so parity and
benchmarks use seeded synthetic weights of exactly these shapes.
"""
import numpy as np

BN_EPS = 1e-3   # tf.contrib.layers.batch_norm default epsilon (tf_util.py:527)


def layer_table(n_max_parts, mixed_pred=True, early_split_nocs=True, scope="SPFN"):
    """[(full scope, cin, cout, has_bn, conv kind)] for every layer on the inference graph."""
    K = n_max_parts
    t = []
    e = scope + "/est_net/"
    for name, cin, mlp in (("layer1", 3, (64, 64, 128)), ("layer2", 131, (128, 128, 256)),
                           ("layer3", 259, (256, 512, 1024))):          # architectures.py:62-75
        for i, c in enumerate(mlp):
            t.append((f"{e}{name}/conv{i}", cin, c, True, "conv2d"))
            cin = c
    for name, cin, mlp in (("fa_layer1", 1280, (256, 256)), ("fa_layer2", 384, (256, 128)),
                           ("fa_layer3", 131, (128, 128, 128))):       # architectures.py:78-86
        for i, c in enumerate(mlp):
            t.append((f"{e}{name}/conv_{i}", cin, c, True, "conv2d"))
            cin = c
    t.append((e + "fc1", 128, 128, True, "conv1d"))                    # architectures.py:89
    out_dims = [K, 3 * K] + ([K, 3 * K] if mixed_pred else []) + [1]   # lib/architecture.py:98-102
    n = scope + "/nocs_net/"
    for i, d in enumerate(out_dims):
        if early_split_nocs and i == 1:
            t.append((f"{n}fc11_{i}", 128, 128, False, "conv1d"))      # lib/architecture.py:111
        t.append((f"{n}fc2_{i}", 128, d, False, "conv1d"))
    j = scope + "/joint_net/"
    t.append((j + "fc3_0", 128, 128, True, "conv1d"))
    t.append((j + "fc3_1", 128, 128, True, "conv1d"))
    for i, d in enumerate((3, 3, 1, 3)):                                # fc4_3 -> n_max_parts=3 default, :195
        t.append((f"{j}fc4_{i}", 128, d, False, "conv1d"))
    return t


def synthetic_weights(n_max_parts, mixed_pred=True, early_split_nocs=True, seed=0, scope="SPFN"):
    """Xavier-uniform kernels, small biases, non-trivial BN statistics (SURVEY 8a "weights")."""
    rng = np.random.RandomState(seed)
    w = {}
    for full, cin, cout, bn, kind in layer_table(n_max_parts, mixed_pred, early_split_nocs, scope):
        lim = np.sqrt(6.0 / (cin + cout))
        shape = (1, 1, cin, cout) if kind == "conv2d" else (1, cin, cout)
        w[full + "/weights"] = rng.uniform(-lim, lim, shape).astype(np.float32)
        w[full + "/biases"] = (0.05 * rng.randn(cout)).astype(np.float32)
        if bn:
            w[full + "/bn/gamma"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            w[full + "/bn/beta"] = (0.1 * rng.randn(cout)).astype(np.float32)
            w[full + "/bn/moving_mean"] = (0.1 * rng.randn(cout)).astype(np.float32)
            w[full + "/bn/moving_variance"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    return w


def fold_layer(weights, full_scope):
    """TF variables of one conv layer -> (w[cin,cout], b, scale, shift) float32.
    Inference BN (tf.nn.batch_normalization): inv = gamma*rsqrt(var+eps); y = x*inv + (beta-mean*inv)."""
    k = np.asarray(weights[full_scope + "/weights"], np.float32)
    w = np.ascontiguousarray(k.reshape(k.shape[-2], k.shape[-1]))
    b = np.asarray(weights[full_scope + "/biases"], np.float32)
    cout = w.shape[1]
    if full_scope + "/bn/gamma" in weights:
        gamma = np.asarray(weights[full_scope + "/bn/gamma"], np.float32)
        beta = np.asarray(weights[full_scope + "/bn/beta"], np.float32)
        mean = np.asarray(weights[full_scope + "/bn/moving_mean"], np.float32)
        var = np.asarray(weights[full_scope + "/bn/moving_variance"], np.float32)
        scale = (gamma * (np.float32(1.0) / np.sqrt(var + np.float32(BN_EPS)))).astype(np.float32)
        shift = (beta - mean * scale).astype(np.float32)
    else:
        scale, shift = np.ones(cout, np.float32), np.zeros(cout, np.float32)
    return dict(w=w, b=b, scale=scale, shift=shift)


def save_npz(path, weights):
    np.savez(path, **{k.replace("/", "__"): v for k, v in weights.items()})


def load_npz(path):
    with np.load(path) as z:
        return {k.replace("__", "/"): z[k] for k in z.files}
