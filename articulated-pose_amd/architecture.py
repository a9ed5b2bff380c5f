"""Inference-only counterpart of lib/architecture.py: get_per_point_model_new (:86-161) and
joint_est_model (:195-208).

The ten tiny head convolutions are issued as three GEMM launches on column-concatenated kernels
(every output column is an independent dot product, so this is exact), writing one logits row per
point; a single kernel then applies softmax / sigmoid / tanh and composes gocs (:124-159).
"""
import torch

from . import _lib, tf_util
from .architectures import build_pointnet2_shared


def joint_est_model(scope, X, is_training, bn_decay, n_max_parts=3, pred_joint_ind=False):
    """Returns the RAW (pre-activation) joint_axis, unitvec, heatmap, joint_cls like the reference."""
    layer_dims = [128, 128]
    with tf_util.variable_scope(scope):
        for j, dim in enumerate(layer_dims):
            X = tf_util.conv1d(X, dim, 1, padding='VALID', bn=True, is_training=is_training,
                               scope='fc3_{}'.format(j), bn_decay=bn_decay)
            X = tf_util.dropout(X, keep_prob=0.5, is_training=is_training, scope='dp1')
        joint_axis = tf_util.conv1d(X, 3, 1, padding='VALID', activation_fn=None, scope='fc4_0')
        univect = tf_util.conv1d(X, 3, 1, padding='VALID', activation_fn=None, scope='fc4_1')
        heatmap = tf_util.conv1d(X, 1, 1, padding='VALID', activation_fn=None, scope='fc4_2')
        joint_cls = tf_util.conv1d(X, n_max_parts, 1, padding='VALID', activation_fn=None, scope='fc4_3')
    return joint_axis, univect, heatmap, joint_cls


def get_per_point_model_new(scope, P, n_max_parts, is_training, bn_decay, early_split=False, early_split_nocs=False,
                            mixed_pred=False, pred_joint=False, pred_joint_ind=False):
    '''
        Inputs:
            - P: BxNx3 tensor, the input point cloud
            - K := n_max_parts
        Outputs: a dict with W (BxNxK softmax), nocs_per_point (BxNx3K), confi_per_point (BxNx1),
            heatmap_per_point, unitvec_per_point, joint_axis_per_point, index_per_point and, when
            mixed_pred, gocs_per_point, global_scale, global_translation.
    '''
    K = n_max_parts
    _lib.require_cuda(P)
    P = P.contiguous().float()
    B, N, _ = P.shape
    rows = B * N
    with tf_util.variable_scope(scope):
        out_dims = [K, 3 * K] + ([K, 3 * K] if mixed_pred else []) + [1]
        net = build_pointnet2_shared('est_net', X=P, out_dims=out_dims, is_training=is_training, bn_decay=bn_decay)
        dev = net.device
        n_head = sum(out_dims)
        ld = (n_head + 10 + 3) // 4 * 4
        logits = torch.empty((rows, ld), dtype=torch.float32, device=dev)

        with tf_util.variable_scope('nocs_net'):
            names = [tf_util.current_scope('fc2_{}'.format(i)) for i in range(len(out_dims))]
            if early_split_nocs:
                # columns of fc2_1 are produced from fc11_1's output below; keep a zero placeholder
                cat = tf_util.get_layer_concat([names[0]] + names[2:], dev, zero_cols={0: 3 * K})
            else:
                cat = tf_util.get_layer_concat(names, dev)
            tf_util.conv_rows(net, rows, 128, 128, cat, False, out=logits, ldy=ld)
            if early_split_nocs:
                l11 = tf_util.get_layer(tf_util.current_scope('fc11_1'), dev)
                shared = tf_util.conv_rows(net, rows, 128, 128, l11, False)     # no BN, no activation (:111)
                l21 = tf_util.get_layer(names[1], dev)
                tf_util.conv_rows(shared, rows, 128, 128, l21, False, out=logits[:, K:], ldy=ld)

        with tf_util.variable_scope('joint_net'):
            X = net
            for j in range(2):
                lay = tf_util.get_layer(tf_util.current_scope('fc3_{}'.format(j)), dev)
                X = tf_util.conv_rows(X, rows, 128, 128, lay, True)
            cat = tf_util.get_layer_concat([tf_util.current_scope('fc4_{}'.format(i)) for i in range(4)], dev)
            tf_util.conv_rows(X, rows, 128, 128, cat, False, out=logits[:, n_head:], ldy=ld)

    def new(c):
        return torch.empty((B, N, c), dtype=torch.float32, device=dev)

    pred = {
        'W': new(K), 'nocs_per_point': new(3 * K), 'confi_per_point': new(1), 'heatmap_per_point': new(1),
        'unitvec_per_point': new(3), 'joint_axis_per_point': new(3), 'index_per_point': new(3),
    }
    if mixed_pred:
        pred['gocs_per_point'] = new(3 * K)
        pred['global_scale'] = new(K)
        pred['global_translation'] = new(3 * K)
    _lib.call("ancsh_head_activations", rows, K, 1 if mixed_pred else 0, _lib.ptr(logits), ld,
              _lib.ptr(pred['W']), _lib.ptr(pred['nocs_per_point']), _lib.ptr(pred['confi_per_point']),
              _lib.ptr(pred['heatmap_per_point']), _lib.ptr(pred['unitvec_per_point']),
              _lib.ptr(pred['joint_axis_per_point']), _lib.ptr(pred['index_per_point']),
              _lib.ptr(pred.get('gocs_per_point')), _lib.ptr(pred.get('global_scale')),
              _lib.ptr(pred.get('global_translation')))
    return pred
