"""The ANCSH and the NPCS network of the evaluation pipeline on ONE batch of clouds, layer by layer in grouped launches.

The reference runs `main.py --nocs_type=ancsh --test` and `main.py --nocs_type=npcs --test` as two separate TensorFlow sessions
(main.py:42-52) over the same test clouds; evaluation/parallel_ancsh_pose.py then reads both prediction files (:232-237).  The
two networks share their backbone's shapes (pointnet_plusplus/architectures.py:62-86) and differ only in weights and heads, so
here every backbone layer of both is ONE launch on stacked activations (network-major: rows of network 0 first): the
set-abstraction levels (ancsh_sa_module_fused*_grouped), the eleven small layers of layer3 / fa_layer1 / fa_layer2
(ancsh_conv1x1_packed_grouped, ancsh_conv1x1_grouped: latency-bound launches, a second network's rows ride along almost for free)
and the two interpolation + concat launches (ancsh_fp_interpolate_concat_ex; sampling, grouping and 3-NN depend only on the cloud
and are computed once).  The tails (fa_layer3 + fc1 + heads: one ancsh_mlp_chain launch per network) differ in their head
structure and stay separate.  Every output is bit-identical to Network.predict of the same network on its own
(tests/test_network_gpu.py::test_paired_networks_equal_separate_forwards).
"""
import ctypes

import torch

from . import _lib, architecture, pointnet_util, tf_util

_VP = ctypes.c_void_p

import os as _os
TAIL_FP = _os.environ.get('ANCSH_TAIL_FP', '1') != '0'          # fa_layer3's interpolation inside the tail chain's tile load; 0 = concat buffer + plain chain, same bits
MID_CHAIN = _os.environ.get('ANCSH_MID_CHAIN', '1') != '0'      # layer3 / fa_layer1 / fa_layer2 as chain launches (csrc/mid_chain.hip); 0 = layer by layer, same bits


class _Table(object):
    """host array of device pointers handed to an ABI call as `const float *const *`; keeps the ctypes array alive"""

    def __init__(self, ptrs):
        self.arr = (_VP * len(ptrs))(*ptrs)
        self.p = ctypes.cast(self.arr, _VP)


def _table(ptrs):
    return _Table(ptrs)


class PairedNetworks(object):
    """nets: Network objects (same n_max_parts, device and scope).  predict(P, geometry=None) -> [pred dict per network]."""

    SA1, SA2, SA3 = (512, 0.2, 64, (64, 64, 128)), (128, 0.4, 64, (128, 128, 256)), (256, 512, 1024)

    def __init__(self, nets):
        self.nets = list(nets)
        n0 = self.nets[0]
        if not 1 <= len(self.nets) <= 4 or any(n.n_max_parts != n0.n_max_parts or n.device != n0.device or n.scope != n0.scope for n in self.nets):
            raise ValueError("PairedNetworks: 1..4 networks with the same n_max_parts, device and scope")
        self.device, self.scope = n0.device, n0.scope

    def eligible(self):
        """True when every network takes the fused paths this class batches (the ANCSH backbone shapes, chain-sized heads)."""
        return (architecture.FUSED_TAIL and pointnet_util.FUSED_SA and pointnet_util.FP_SINGLE_SOURCE and
                all(architecture._head_dims(n.n_max_parts, n.is_mixed, n.early_split_nocs)[1] for n in self.nets) and self._backbone_shapes_ok())

    def _backbone_shapes_ok(self):
        """The grouped launches below hard-code the ANCSH backbone widths (SA1 / SA2 / SA3 above, fa 1280 -> 256 -> 256, 384 -> 256 ->
        128) and read PACKED weights of exactly those shapes: a store with other widths must take Network.predict, not read a
        packed buffer out of bounds.  Checked once per object against weights.layer_table."""
        if getattr(self, "_shapes_ok", None) is None:
            from .weights import layer_table
            ok = True
            for n in self.nets:
                for full, cin, cout, _bn, _kind in layer_table(n.n_max_parts, n.is_mixed, n.early_split_nocs, self.scope):
                    w = n.weights.get(full + "/weights")
                    ok = ok and w is not None and tuple(w.shape[-2:]) == (cin, cout)
            self._shapes_ok = bool(ok)
        return self._shapes_ok

    # ---- parameters ----------------------------------------------------------------------------------------------------------
    def _layers(self, rel_scope):
        out = []
        for net in self.nets:
            tf_util.set_variables(net.weights)
            out.append(tf_util.get_layer(self.scope + "/est_net/" + rel_scope, self.device))
        return out

    @staticmethod
    def _ep_tables(layers):
        return [_table([_lib.ptr(l[k]) for l in layers]) for k in ("b", "scale", "shift")]

    def _conv(self, layers, x, rows, cin, ldx, cout, relu=True, pool=0, acc_init=None, init_rows=0, row0=0, raw=False):
        """one grouped layer launch: x (G * rows, ldx) -> (G * rows [/ pool], cout); the kernel rows [row0:] of every network"""
        G = len(self.nets)
        y = torch.empty((G * (rows // pool if pool else rows), cout), dtype=torch.float32, device=self.device)
        w = _table([_lib.ptr(tf_util.packed_weight(l, row0)) for l in layers])
        b, sc, sh = (None, None, None) if raw else self._ep_tables(layers)
        _lib.call("ancsh_conv1x1_packed_grouped", G, rows, cin, cout, _lib.ptr(x), ldx, w.p, b and b.p, sc and sc.p, sh and sh.p,
                  2 if raw else (1 if relu else 0), _lib.ptr(y), cout, pool, _lib.ptr(acc_init), init_rows)
        return y

    # ---- the middle of the backbone: layer3, fa_layer1, fa_layer2 ----------------------------------------------------------------
    def _mid_layers(self, B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2):
        """layer by layer (rounds 3-4; ANCSH_MID_CHAIN=0): nine conv launches, a concat and an interpolate + concat launch"""
        G, dev = len(self.nets), self.device
        f = dict(dtype=torch.float32, device=dev)
        # layer3: the whole level-2 cloud as one neighbourhood (group_all): rows [xyz | features], 259 -> 256 -> 512 -> 1024 + max
        x3 = torch.cat([l2_xyz.unsqueeze(0).expand(G, B, 128, 3).reshape(G * B, 128, 3), l2_points], dim=2)       # (G*B, 128, 259)
        h = self._conv(L3[0], x3, B * 128, 259, 259, 256)
        h = self._conv(L3[1], h, B * 128, 256, 256, 512)
        l3_points = self._conv(L3[2], h, B * 128, 512, 512, 1024, pool=128)                                       # (G*B, 1024)

        # fa_layer1: the interpolation source is one point per cloud -> its share of the first dot product once per cloud
        init = torch.empty((G * B, 256), **f)
        w1 = _table([_lib.ptr(l["w"]) for l in F1[0]])
        _lib.call("ancsh_conv1x1_grouped", G, B, 1024, 256, _lib.ptr(l3_points), 1024, w1.p, None, None, None, 2, _lib.ptr(init), 256, 0)
        h = self._conv(F1[0], l2_points, B * 128, 256, 256, 256, acc_init=init, init_rows=128, row0=1024)
        l2_up = self._conv(F1[1], h, B * 128, 256, 256, 256)                                                      # (G*B*128, 256)

        # fa_layer2: [interpolated level-2 features (256) | level-1 features (128)] -> 256 -> 128
        buf = torch.empty((G * B, 512, 384), **f)
        _lib.call("ancsh_fp_interpolate_concat_ex", G * B, 128, 256, 512, _lib.ptr(l2_up), _lib.ptr(fi2), _lib.ptr(fw2), _lib.ptr(l1_points), 128,
                  _lib.ptr(buf), 384, B, G * B)
        h = self._conv(F2[0], buf, B * 512, 384, 384, 256)
        l1_up = self._conv(F2[1], h, B * 512, 256, 256, 128)                                                      # (G*B*512, 128)
        return l1_up

    def _mid_chains_split16(self, B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2):
        """the three levels on the 16-bit matrix pipe (opt-in experiment, level 4: csrc/mid_bf16x3.hip); layer3 keeps the f32 chain where the
        scheme's planes of a 64 x 512 tile do not fit the LDS (bf16x3)"""
        G, dev = len(self.nets), self.device
        f = dict(dtype=torch.float32, device=dev)
        name = pointnet_util.split_name

        def params(layers_per_level, row0s):
            return _table([_lib.ptr(v) for g in range(G) for ls, r0 in zip(layers_per_level, row0s)
                           for v in (pointnet_util._bf16x3_weight(ls[g], r0), ls[g]["b"], ls[g]["scale"], ls[g]["shift"])])

        npts = l2_points.shape[1]
        if pointnet_util.SPLIT_SCHEME == "f16x2":
            p3 = params(L3, (0, 0, 0))
            nparts = npts // 64
            tile_max = torch.empty((G * B, nparts, 1024), **f)
            _lib.call(name("ancsh_sa3_chain_grouped_bf16x3"), G, B, npts, 256, 256, 512, 1024, _lib.ptr(l2_xyz), _lib.ptr(l2_points), p3.p, _lib.ptr(tile_max))
        else:
            p3 = _table([_lib.ptr(v) for g in range(G) for ls in L3 for v in (tf_util.packed_weight(ls[g], 0), ls[g]["b"], ls[g]["scale"], ls[g]["shift"])])
            nparts = npts // 32
            tile_max = torch.empty((G * B, nparts, 1024), **f)
            _lib.call("ancsh_sa3_chain_grouped", G, B, npts, 256, 256, 512, 1024, _lib.ptr(l2_xyz), _lib.ptr(l2_points), p3.p, _lib.ptr(tile_max))
        init = torch.empty((G * B, 256), **f)
        w1 = _table([_lib.ptr(l["w"]) for l in F1[0]])
        _lib.call("ancsh_fp_single_source_init", G, B, 1024, 256, nparts, _lib.ptr(tile_max), w1.p, _lib.ptr(init))      # exact f32 chain (VALU), as in the f32 path
        p1 = params(F1, (1024, 0))
        l2_up = torch.empty((G * B * npts, 256), **f)
        _lib.call(name("ancsh_fp1_chain_grouped_bf16x3"), G, B, npts, 256, 256, 256, _lib.ptr(l2_points), _lib.ptr(init), p1.p, _lib.ptr(l2_up))
        n1 = l1_points.shape[1]
        p2 = params(F2, (0, 0))
        l1_up = torch.empty((G * B * n1, 128), **f)
        _lib.call(name("ancsh_fp2_chain_grouped_bf16x3"), G, B, npts, n1, 256, 128, 256, 128, _lib.ptr(l2_up), _lib.ptr(fi2), _lib.ptr(fw2), _lib.ptr(l1_points),
                  p2.p, _lib.ptr(l1_up))
        return l1_up

    def _mid_chains(self, B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2):
        """the same three levels as chain launches (csrc/mid_chain.hip): activations stay in LDS, bit-identical outputs"""
        G, dev = len(self.nets), self.device
        f = dict(dtype=torch.float32, device=dev)

        def params(layers_per_level, row0s):
            return _table([_lib.ptr(v) for g in range(G) for ls, r0 in zip(layers_per_level, row0s)
                           for v in (tf_util.packed_weight(ls[g], r0), ls[g]["b"], ls[g]["scale"], ls[g]["shift"])])

        # layer3: rows [xyz | features] built in the kernel's load; out = the maxima of every 32-row tile
        npts = l2_points.shape[1]
        p3 = params(L3, (0, 0, 0))
        tile_max = torch.empty((G * B, npts // 32, 1024), **f)
        _lib.call("ancsh_sa3_chain_grouped", G, B, npts, 256, 256, 512, 1024, _lib.ptr(l2_xyz), _lib.ptr(l2_points), p3.p, _lib.ptr(tile_max))
        # fa_layer1: the single-source share of its first dot product once per cloud (the maximum over the tiles taken in the load) ...
        init = torch.empty((G * B, 256), **f)
        w1 = _table([_lib.ptr(l["w"]) for l in F1[0]])
        _lib.call("ancsh_fp_single_source_init", G, B, 1024, 256, npts // 32, _lib.ptr(tile_max), w1.p, _lib.ptr(init))
        # ... then both layers on the level-2 points
        p1 = params(F1, (1024, 0))
        l2_up = torch.empty((G * B * npts, 256), **f)
        _lib.call("ancsh_fp1_chain_grouped", G, B, npts, 256, 256, 256, _lib.ptr(l2_points), _lib.ptr(init), p1.p, _lib.ptr(l2_up))
        # fa_layer2: interpolation from the three nearest level-2 points in the kernel's load, then 384 -> 256 -> 128
        n1 = l1_points.shape[1]
        p2 = params(F2, (0, 0))
        l1_up = torch.empty((G * B * n1, 128), **f)
        _lib.call("ancsh_fp2_chain_grouped", G, B, npts, n1, 256, 128, 256, 128, _lib.ptr(l2_up), _lib.ptr(fi2), _lib.ptr(fw2), _lib.ptr(l1_points),
                  p2.p, _lib.ptr(l1_up))
        return l1_up

    # ---- forward -------------------------------------------------------------------------------------------------------------
    def predict(self, P, geometry=None):
        if not torch.is_tensor(P):
            import numpy as np
            P = torch.from_numpy(np.ascontiguousarray(P, np.float32))
        P = P.to(self.device).contiguous().float()
        _lib.require_cuda(P)
        if not self.eligible():
            return [n.predict(P, geometry) for n in self.nets]
        G, (B, N, _), dev = len(self.nets), P.shape, self.device
        est = self.scope + "/est_net/"
        geo = geometry if geometry is not None else pointnet_util.Geometry()
        k1 = ("sa", est + "layer1", 512, 0.2, 64)
        if k1 not in geo:
            pointnet_util.precompute_geometry(P, geo, scope=self.scope + "/est_net")
        (l1_xyz, idx1), (l2_xyz, idx2) = geo[k1], geo[("sa", est + "layer2", 128, 0.4, 64)]
        (fi2, fw2), (fi3, fw3) = geo[("fp", est + "fa_layer2")], geo[("fp", est + "fa_layer3")]
        f = dict(dtype=torch.float32, device=dev)

        # layer1: 3 -> 64 -> 64 -> 128 on every neighbourhood of every network, one launch
        L1 = [self._layers("layer1/conv%d" % i) for i in range(3)]
        for ls in L1:
            for l in ls:
                tf_util.packed_weight(l)
        l1_points = torch.empty((G * B, 512, 128), **f)
        bx3 = pointnet_util.SA_BF16X3        # opt-in: the SA levels on the bf16 matrix pipe (six bf16 products per f32 product, csrc/sa_bf16x3.hip)
        if bx3 >= 1:
            p1 = _table([_lib.ptr(v) for g in range(G) for i in range(3)
                         for v in (pointnet_util._bf16x3_weight(L1[i][g]), L1[i][g]["b"], L1[i][g]["scale"], L1[i][g]["shift"])])
            _lib.call(pointnet_util.split_name("ancsh_sa_module_fused_bf16x3_grouped"), G, B, N, 512, 64, 0, 64, 64, 128, _lib.ptr(P), None, _lib.ptr(l1_xyz),
                      _lib.ptr(idx1), p1.p, _lib.ptr(l1_points))
        else:
            p1 = _table([_lib.ptr(L1[i][g][k]) for g in range(G) for i in range(3) for k in ("w_packed", "b", "scale", "shift")])
            _lib.call("ancsh_sa_module_fused_grouped", G, B, N, 512, 64, 0, 64, 64, 128, _lib.ptr(P), None, _lib.ptr(l1_xyz), _lib.ptr(idx1),
                      p1.p, _lib.ptr(l1_points))

        # layer2: the first layer's feature part once per level-1 point (raw partial sums), then the fused level
        L2 = [self._layers("layer2/conv%d" % i) for i in range(3)]
        first = [tf_util.sa_first_layer_split(l) for l in L2[0]]
        for ls in L2[1:]:
            for l in ls:
                tf_util.packed_weight(l)
        partial = self._conv(L2[0], l1_points, B * 512, 128, 128, 128, raw=True, row0=3)
        l2_points = torch.empty((G * B, 128, 256), **f)
        if bx3 >= 2:
            p2 = _table([_lib.ptr(v) for g in range(G) for v in
                         ([pointnet_util._bf16x3_xyz_weight(first[g], 128), first[g]["b"], first[g]["scale"], first[g]["shift"]] +
                          [x for i in (1, 2) for x in (pointnet_util._bf16x3_weight(L2[i][g]), L2[i][g]["b"], L2[i][g]["scale"], L2[i][g]["shift"])])])
            _lib.call(pointnet_util.split_name("ancsh_sa_module_fused_partial_bf16x3_grouped"), G, B, 512, 128, 64, 128, 128, 256, _lib.ptr(l1_xyz), _lib.ptr(partial),
                      _lib.ptr(l2_xyz), _lib.ptr(idx2), p2.p, _lib.ptr(l2_points))
        else:
            p2 = _table([_lib.ptr(v) for g in range(G) for v in
                         ([first[g]["w_xyz_packed"], first[g]["b"], first[g]["scale"], first[g]["shift"]] +
                          [L2[i][g][k] for i in (1, 2) for k in ("w_packed", "b", "scale", "shift")])])
            _lib.call("ancsh_sa_module_fused_partial_grouped", G, B, 512, 128, 64, 128, 128, 256, _lib.ptr(l1_xyz), _lib.ptr(partial),
                      _lib.ptr(l2_xyz), _lib.ptr(idx2), p2.p, _lib.ptr(l2_points))

        L3 = [self._layers("layer3/conv%d" % i) for i in range(3)]
        F1 = [self._layers("fa_layer1/conv_%d" % i) for i in range(2)]
        F2 = [self._layers("fa_layer2/conv_%d" % i) for i in range(2)]
        if bx3 >= 4 and l2_points.shape[1] % 64 == 0 and l1_points.shape[1] % 64 == 0:
            l1_up = self._mid_chains_split16(B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2)
        elif MID_CHAIN:
            l1_up = self._mid_chains(B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2)
        else:
            l1_up = self._mid_layers(B, l2_xyz, l2_points, l1_points, fi2, fw2, L3, F1, F2)

        tail_bx3 = bx3 >= 3 and N % 64 == 0       # opt-in: the tail chain on the bf16 matrix pipe too (csrc/tail_bf16x3.hip)
        progs = []
        for net in self.nets:
            tf_util.set_variables(net.weights)
            with tf_util.variable_scope(self.scope):
                progs.append(architecture._tail_program(B * N, net.n_max_parts, net.is_mixed, net.early_split_nocs, dev, bf16x3=tail_bx3))
        if tail_bx3:
            architecture.run_tail_programs_bf16x3(progs, (B, N, 512, l1_up.view(G * B, 512, 128), fi3, fw3, P))
        elif TAIL_FP and N % 128 == 0:
            # fa_layer3's input rows [interpolated (128) | xyz (3)] are built in the chain's tile load: both networks' chains in ONE launch
            # (two waves per SIMD), no (G * B, N, 132) concat buffer written and read back, no interpolate + concat launch
            architecture.run_tail_programs(None, B * N, progs, fp=(B, N, 512, l1_up.view(G * B, 512, 128), fi3, fw3, P))
        else:
            # ... or materialised for every network in one launch (ragged N: a chain workgroup's four tiles must stay inside a cloud)
            x = torch.empty((G * B, N, 132), **f)
            _lib.call("ancsh_fp_interpolate_concat_ex", G * B, 512, 128, N, _lib.ptr(l1_up), _lib.ptr(fi3), _lib.ptr(fw3), _lib.ptr(P), 3,
                      _lib.ptr(x), 132, B, B)
            architecture.run_tail_programs(x, B * N, progs)
        return [architecture._activations(p[2], p[3], B, N, net.n_max_parts, net.is_mixed) for p, net in zip(progs, self.nets)]
