"""CPU ORACLE of the ANCSH / NPCS network forward (test infrastructure only).

Restates, on numpy arrays over the C routines of ancsh_oracle.c, the inference graph
    lib/architecture.py:86-161 (get_per_point_model_new), :195-208 (joint_est_model)
    pointnet_plusplus/architectures.py:56-95 (build_pointnet2_shared)
    pointnet_plusplus/utils/pointnet_util.py:29-91 (sample_and_group[_all]), :94-161 (SA), :206-236 (FP)
op by op, with NO fusion (grouped tensors, concats and per-layer activations are materialised the
way the TF graph does).  TensorFlow's convolution arithmetic is a third-party dependency absent from /root/reference: the
summation order of a dot product is this restatement's choice (k ascending; the first layer of a grouped level with input
features sums the feature channels before the three centred coordinates, see sa_module).  Weights: dict keyed by TF variable names (same layout the product reads).
"""
import numpy as np

from . import oracle as O

BN_EPS = 1e-3   # tf.contrib.layers.batch_norm default (tf_util.py:527)


def fold(weights, scope):
    """TF variables -> (w[cin,cout], b, scale, shift): y = (x.w + b)*scale + shift
    with scale = gamma*rsqrt(var+eps), shift = beta - mean*scale (tf.nn.batch_normalization)."""
    k = np.asarray(weights[scope + "/weights"], np.float32)
    w = np.ascontiguousarray(k.reshape(k.shape[-2], k.shape[-1]))
    b = np.asarray(weights[scope + "/biases"], np.float32)
    if scope + "/bn/gamma" in weights:
        g = np.asarray(weights[scope + "/bn/gamma"], np.float32)
        be = np.asarray(weights[scope + "/bn/beta"], np.float32)
        mu = np.asarray(weights[scope + "/bn/moving_mean"], np.float32)
        var = np.asarray(weights[scope + "/bn/moving_variance"], np.float32)
        scale = (g * (np.float32(1.0) / np.sqrt(var + np.float32(BN_EPS)))).astype(np.float32)
        shift = (be - mu * scale).astype(np.float32)
    else:
        scale, shift = np.ones(w.shape[1], np.float32), np.zeros(w.shape[1], np.float32)
    return dict(w=w, b=b, scale=scale, shift=shift)


def conv(weights, scope, x, act=True):
    return O.conv1x1(x, fold(weights, scope), 1 if act else 0)


def sample_and_group(npoint, radius, nsample, xyz, points):            # pointnet_util.py:29-63
    idx_fps = O.farthest_point_sample(npoint, xyz)
    new_xyz = O.gather_point(xyz, idx_fps)
    idx, cnt = O.query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = O.group_point(xyz, idx)
    grouped_xyz = grouped_xyz - new_xyz[:, :, None, :]                  # tf.tile + subtract (:53)
    if points is not None and points.shape[2] > 0:
        new_points = np.concatenate([grouped_xyz, O.group_point(points, idx)], axis=-1)
    else:
        new_points = grouped_xyz                                        # concat with a 0-channel tensor
    return new_xyz, np.ascontiguousarray(new_points, np.float32), idx, idx_fps


def sa_module(weights, scope, xyz, points, npoint, radius, nsample, mlp, group_all):   # :94-161
    aux = {}
    if group_all:
        b, n, _ = xyz.shape
        new_xyz = np.zeros((b, 1, 3), np.float32)
        new_points = np.concatenate([xyz, points], axis=2)[:, None]     # (b,1,n,3+c) (:84-87)
    else:
        new_xyz, new_points, idx, idx_fps = sample_and_group(npoint, radius, nsample, xyz, points)
        aux = dict(idx=idx, fps=idx_fps)
    x = new_points
    for i, _c in enumerate(mlp):
        if i == 0 and not group_all and points is not None and points.shape[2] > 0:
            # tf.nn.conv2d does not define a summation order.  The restatement fixes one: k ascending, except that the first layer of
            # a grouped level with input features ([x_j - c | f_j], pointnet_util.py:55) sums the FEATURE channels first and the three
            # centred coordinates last (the feature part is then the same in every neighbourhood a point falls into).
            f = fold(weights, f"{scope}/conv{i}")
            f["w"] = np.ascontiguousarray(np.concatenate([f["w"][3:], f["w"][:3]], axis=0))
            x = O.conv1x1(np.ascontiguousarray(np.concatenate([x[..., 3:], x[..., :3]], axis=-1)), f, 1)
            continue
        x = conv(weights, f"{scope}/conv{i}", x)
    x = O.group_max(x)                                                  # reduce_max over nsample (:134)
    return new_xyz, x, aux


def fp_module(weights, scope, xyz1, xyz2, points1, points2, mlp):       # :206-236
    dist, idx = O.three_nn(xyz1, xyz2)
    weight = O.three_weights(dist)
    interp = O.three_interpolate(points2, idx, weight)
    x = np.concatenate([interp, points1], axis=2) if points1 is not None else interp
    for i, _c in enumerate(mlp):
        x = conv(weights, f"{scope}/conv_{i}", np.ascontiguousarray(x, np.float32))
    return x


def forward(weights, P, n_max_parts, mixed_pred=True, early_split_nocs=True, scope="SPFN", return_aux=False):
    """P (B,N,3) float32 -> dict of the reference's pred_dict tensors (lib/architecture.py:141-159)."""
    P = np.ascontiguousarray(P, np.float32)
    K = n_max_parts
    e = scope + "/est_net"
    l0_xyz = P
    l1_xyz, l1_points, a1 = sa_module(weights, e + "/layer1", l0_xyz, None, 512, 0.2, 64, (64, 64, 128), False)
    l2_xyz, l2_points, a2 = sa_module(weights, e + "/layer2", l1_xyz, l1_points, 128, 0.4, 64, (128, 128, 256), False)
    l3_xyz, l3_points, _ = sa_module(weights, e + "/layer3", l2_xyz, l2_points, None, None, None, (256, 512, 1024), True)
    l2_points_fp = fp_module(weights, e + "/fa_layer1", l2_xyz, l3_xyz, l2_points, l3_points, (256, 256))
    l1_points_fp = fp_module(weights, e + "/fa_layer2", l1_xyz, l2_xyz, l1_points, l2_points_fp, (256, 128))
    l0_points = fp_module(weights, e + "/fa_layer3", l0_xyz, l1_xyz, l0_xyz, l1_points_fp, (128, 128, 128))
    net = conv(weights, e + "/fc1", l0_points)                          # + dropout = identity at test

    out_dims = [K, 3 * K] + ([K, 3 * K] if mixed_pred else []) + [1]
    res = []
    for i, _d in enumerate(out_dims):                                   # lib/architecture.py:105-120
        shared = net
        if early_split_nocs and i == 1:
            shared = conv(weights, f"{scope}/nocs_net/fc11_{i}", shared, act=False)
        res.append(conv(weights, f"{scope}/nocs_net/fc2_{i}", shared, act=False))
    if mixed_pred:
        W, nocs, scale, trans, confi = res
        scale = O.activation(scale, "sigmoid")
        trans = O.activation(trans, "tanh")
    else:
        W, nocs, confi = res
    X = net
    for j in range(2):                                                  # joint_est_model :195-208
        X = conv(weights, f"{scope}/joint_net/fc3_{j}", X)
    axis = conv(weights, f"{scope}/joint_net/fc4_0", X, act=False)
    unitvec = conv(weights, f"{scope}/joint_net/fc4_1", X, act=False)
    heatmap = conv(weights, f"{scope}/joint_net/fc4_2", X, act=False)
    joint_cls = conv(weights, f"{scope}/joint_net/fc4_3", X, act=False)

    pred = {
        "W": O.activation(W, "softmax"),
        "nocs_per_point": O.activation(nocs, "sigmoid"),
        "confi_per_point": O.activation(confi, "sigmoid"),
        "heatmap_per_point": O.activation(heatmap, "sigmoid"),
        "unitvec_per_point": O.activation(unitvec, "tanh"),
        "joint_axis_per_point": O.activation(axis, "tanh"),
        "index_per_point": O.activation(joint_cls, "softmax"),
    }
    if mixed_pred:
        tiled = np.repeat(scale, 3, axis=2)                             # (:154) expand_dims/tile/reshape
        pred["gocs_per_point"] = (pred["nocs_per_point"] * tiled + trans).astype(np.float32)
        pred["global_scale"] = scale
        pred["global_translation"] = trans
    if return_aux:
        pred["_aux"] = dict(l1_xyz=l1_xyz, l2_xyz=l2_xyz, l1_points=l1_points, l2_points=l2_points,
                            l3_points=l3_points, net=net, fps1=a1["fps"], fps2=a2["fps"], idx1=a1["idx"], idx2=a2["idx"])
    return pred
