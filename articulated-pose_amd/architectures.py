"""The PointNet++ backbone both ANCSH networks share: inference-only counterpart of
pointnet_plusplus/architectures.py::build_pointnet2_shared (:56-95), written as a level table.

Levels (what the reference hard-codes call by call):
    set abstraction   layer1: 512 centroids, r = 0.2, 64 samples, MLP 64-64-128
                      layer2: 128 centroids, r = 0.4, 64 samples, MLP 128-128-256
                      layer3: the whole cloud as one group,        MLP 256-512-1024
    feature propagation (coarse -> fine, skip connections)
                      fa_layer1 -> level 2: MLP 256-256
                      fa_layer2 -> level 1: MLP 256-128
                      fa_layer3 -> level 0: MLP 128-128-128, skip features = [xyz | input features]
    head trunk        fc1: 128 channels + BN + ReLU, dropout 0.5 (identity at inference)
Variable scopes ('layer1', ..., 'fa_layer3', 'fc1') are the reference's, so its checkpoints resolve."""
import torch

from . import tf_util
from .pointnet_util import pointnet_fp_module, pointnet_sa_module

# (scope, npoint, radius, nsample, mlp, group_all)
SA_LEVELS = (("layer1", 512, 0.2, 64, (64, 64, 128), False),
             ("layer2", 128, 0.4, 64, (128, 128, 256), False),
             ("layer3", None, None, None, (256, 512, 1024), True))
# (scope, mlp) from the coarsest level down
FP_LEVELS = (("fa_layer1", (256, 256)), ("fa_layer2", (256, 128)), ("fa_layer3", (128, 128, 128)))
TRUNK_WIDTH = 128


def build_pointnet2_shared(scope, X, out_dims, is_training, bn_decay):
    """X (B, N, 3 + C) -> per-point trunk features (B, N, 128).  `out_dims` is accepted for signature parity; the heads are
    built by the caller (lib/architecture.py)."""
    with tf_util.variable_scope(scope):
        xyz = [X[:, :, 0:3].contiguous()]
        feats = [X[:, :, 3:]]                      # zero channels for a bare xyz cloud, like tf.slice(X, [0,0,3], [-1,-1,0])
        for name, npoint, radius, nsample, mlp, group_all in SA_LEVELS:
            new_xyz, new_feats, _ = pointnet_sa_module(xyz[-1], feats[-1], npoint=npoint, radius=radius, nsample=nsample,
                                                       mlp=list(mlp), mlp2=None, group_all=group_all, is_training=is_training,
                                                       bn_decay=bn_decay, scope=name)
            xyz.append(new_xyz)
            feats.append(new_feats)
        up = feats[-1]
        for depth, (name, mlp) in enumerate(FP_LEVELS):
            fine = len(SA_LEVELS) - 1 - depth          # level the features are propagated to
            skip = feats[fine] if fine > 0 else torch.cat([xyz[0], feats[0]], dim=-1)
            up = pointnet_fp_module(xyz[fine], xyz[fine + 1], skip, up, list(mlp), is_training, bn_decay, scope=name)
        net = tf_util.conv1d(up, TRUNK_WIDTH, 1, padding='VALID', bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
        return tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp1')
