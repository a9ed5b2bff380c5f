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

    def __init__(self, n_max_parts, weights, nocs_type='ancsh', device='cuda:0', scope='SPFN'):
        self.n_max_parts = n_max_parts
        self.is_mixed = nocs_type == 'ancsh'            # lib/network.py:36-39
        self.early_split_nocs = nocs_type == 'ancsh'    # main.py:45-49
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

    def predict_and_save(self, dset, save_dir, nn_name='SPFN'):
        """dset: iterable of batch dicts with 'P' and (optionally) the GT fields + 'basename_list'."""
        from . import prediction_io
        os.makedirs(save_dir, exist_ok=True)
        n = 0
        for batch in dset:
            pred = {k: v.cpu().numpy() for k, v in self.predict(batch['P']).items()}
            prediction_io.save_batch_nn(nn_name, pred, batch, batch['basename_list'], save_dir,
                                        is_mixed=self.is_mixed, W_reduced=False)
            n += len(batch['basename_list'])
        return n


class AncshEngine(object):
    """Fixed-shape, graph-captured forward: engine = AncshEngine(net, B, N); out = engine(P)."""

    def __init__(self, net, batch_size, num_points, use_graph=True):
        self.net = net
        self.P = torch.zeros((batch_size, num_points, 3), dtype=torch.float32, device=net.device)
        self.graph = None
        self.out = None
        self.stream = torch.cuda.Stream(device=net.device)
        with torch.cuda.stream(self.stream):
            for _ in range(2):                       # warm-up: folds + uploads weights, fills allocator
                self.out = net.predict(self.P)
        self.stream.synchronize()
        if use_graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.out = net.predict(self.P)

    def __call__(self, P=None):
        if P is not None:
            self.P.copy_(P, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.out = self.net.predict(self.P)
        return self.out
