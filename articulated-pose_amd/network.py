"""Inference side of lib/network.py: Network(...).predict / predict_and_save (:257-316) on the MI355X.

`Network` builds nothing ahead of time (there is no graph compiler): a forward is ~45 asynchronous
kernel launches on one HIP stream.  `AncshEngine` pins shapes, captures that launch sequence once
into a hipGraph (torch.cuda.CUDAGraph over torch's caching allocator) and replays it per batch, so
the steady state has no host-side launch cost.
"""
import os

import numpy as np
import torch

from . import architecture, tf_util

PRED_KEYS = ('W', 'nocs_per_point', 'confi_per_point', 'heatmap_per_point', 'unitvec_per_point',
             'joint_axis_per_point', 'index_per_point', 'gocs_per_point', 'global_scale', 'global_translation')


class Network(object):
    """n_max_parts: K.  nocs_type: 'ancsh' (mixed part+global NOCS heads, early-split NOCS branch;
    main.py:42-49) or 'npcs'.  weights: {TF variable name: ndarray} (weights.py)."""

    def __init__(self, n_max_parts, weights, nocs_type='ancsh', device='cuda:0', scope='SPFN',
                 pred_joint=None, pred_joint_ind=None, early_split=None):
        self.n_max_parts = n_max_parts
        self.is_mixed = nocs_type == 'ancsh'            # lib/network.py:36-39
        self.early_split_nocs = nocs_type == 'ancsh'    # main.py:45-49
        # main.py:31-34,42-52: the three flags are argparse store_true options (default False) that only the 'ancsh' branch switches
        # on.  They do not change the graph that is evaluated (joint_est_model is built either way, lib/architecture.py:129) but
        # they select the terms of total_loss (lib/network.py:162-169) and the fields of test_loss.txt (:228-243).
        self.pred_joint = (nocs_type == 'ancsh') if pred_joint is None else bool(pred_joint)
        self.pred_joint_ind = (nocs_type == 'ancsh') if pred_joint_ind is None else bool(pred_joint_ind)
        self.early_split = (nocs_type == 'ancsh') if early_split is None else bool(early_split)
        self.weights = weights
        self.device = torch.device(device)
        self.scope = scope

    def predict(self, P, geometry=None):
        """P: (B,N,3) float32 tensor/ndarray -> dict of device tensors (the reference's pred_dict).
        geometry: optional pointnet_util.Geometry -- empty: filled with this forward's sampling / grouping /
        3-NN results; non-empty (from another network's forward on the SAME P): reused instead of recomputed."""
        if not torch.is_tensor(P):
            P = torch.from_numpy(np.ascontiguousarray(P, np.float32))
        P = P.to(self.device)
        if tf_util._state["weights"] is not self.weights:
            tf_util.set_variables(self.weights)
        from . import pointnet_util
        pointnet_util.use_geometry(geometry)
        try:
            return architecture.get_per_point_model_new(
                scope=self.scope, P=P, n_max_parts=self.n_max_parts, is_training=False, bn_decay=None,
                mixed_pred=self.is_mixed, pred_joint=True, pred_joint_ind=True,
                early_split=self.early_split_nocs, early_split_nocs=self.early_split_nocs)
        finally:
            pointnet_util.use_geometry(None)

    def predict_grouped(self, P, geometry=None):
        """The same forward through the grouped / chained launches (paired.PairedNetworks with this one network: fused SA levels, one
        chain launch per mid-section level, the tail chain with fa_layer3's interpolation in its load: 15 launches instead of ~45)
        when the backbone has the shapes they serve, else predict().  Bit-identical to predict() (tests/test_network_gpu.py)."""
        if getattr(self, "_grouped", None) is None:
            from .paired import PairedNetworks
            one = PairedNetworks([self])
            self._grouped = one if one.eligible() else False
        return self._grouped.predict(P, geometry)[0] if self._grouped else self.predict(P, geometry)

    GT_KEYS = ('nocs_gt', 'cls_gt', 'mask_array', 'heatmap_gt', 'unitvec_gt', 'orient_gt', 'joint_cls_gt', 'joint_cls_mask')

    def predict_and_save(self, dset, save_dir, nn_name='SPFN', coord_regress_loss='L2'):
        """dset: iterable of batch dicts with 'P', 'basename_list' and (optionally) the ground-truth fields of
        fill_gt_dict_with_batch_data (lib/network.py:373-390).  Writes one record per cloud and -- when the ground truth is
        present -- test_loss.txt with the data-weighted mean losses, like lib/network.py:257-316.  Returns
        {'n': records written, 'losses': dict or None, 'msg': the test_loss.txt line or None}."""
        from . import loss as loss_mod
        from . import prediction_io
        os.makedirs(save_dir, exist_ok=True)
        n, n_loss, sums = 0, 0, {}
        need = self.GT_KEYS + (('nocs_gt_g',) if self.is_mixed else ())
        for batch in dset:
            pred_dev = self.predict_grouped(batch['P'])
            size = len(batch['basename_list'])
            if all(k in batch for k in need):
                ld = loss_mod.compute_loss(pred_dev, batch, self.n_max_parts, self.is_mixed, coord_regress_loss)
                for k, v in loss_mod.collect_losses(ld, self.is_mixed, self.pred_joint, self.pred_joint_ind).items():
                    sums[k] = sums.get(k, 0.0) + v * size                    # losses[key] += loss_result[key] * last_step_size
                n_loss += size
            pred = {k: v.cpu().numpy() for k, v in pred_dev.items()}
            prediction_io.save_batch_nn(nn_name, pred, batch, batch['basename_list'], save_dir,
                                        is_mixed=self.is_mixed, W_reduced=False)
            n += size
        losses = msg = None
        if n_loss:
            losses = {k: v / n_loss for k, v in sums.items()}
            # lib/network.py:258-273: only the keys the flags select are accumulated and reported
            msg = loss_mod.format_loss_result(losses, self.is_mixed, pred_joint=self.pred_joint, early_split=self.early_split,
                                              pred_joint_ind=self.pred_joint_ind)
            losses = {k: losses[k] for k in loss_mod.reported_keys(self.is_mixed, self.pred_joint, self.early_split, self.pred_joint_ind)}
            with open(os.path.join(save_dir, 'test_loss.txt'), 'w') as f:
                f.write(msg)
        return {'n': n, 'losses': losses, 'msg': msg}


class AncshEngine(object):
    """Fixed-shape, graph-captured forward: engine = AncshEngine(net, B, N); out = engine(P)."""

    def __init__(self, net, batch_size, num_points, use_graph=True):
        self.net = net
        # the grouped launches with ONE network (paired.PairedNetworks: fused SA levels, the mid-section chains of round 5, the
        # one-tile tail chain) when the backbone has the shapes they serve; bit-identical to net.predict (tests/test_network_gpu.py)
        self._forward = net.predict_grouped
        self.P = torch.zeros((batch_size, num_points, 3), dtype=torch.float32, device=net.device)
        self.graph = None
        self.out = None
        self.stream = torch.cuda.Stream(device=net.device)
        with torch.cuda.stream(self.stream):
            for _ in range(2):                       # warm-up: folds + uploads weights, fills allocator
                self.out = self._forward(self.P)
        self.stream.synchronize()
        if use_graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.out = self._forward(self.P)

    def __call__(self, P=None):
        if P is not None:
            self.P.copy_(P, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.out = self._forward(self.P)
        return self.out
