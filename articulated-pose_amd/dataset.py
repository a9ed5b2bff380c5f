"""The sampling step in front of the network, batched on the MI355X: counterpart of
lib/dataset.py::Dataset.create_unit_data_from_hdf5 (:262-432, the part after the .h5 frame has been parsed into
per-part arrays) and of the batch assembly in Dataset.__next__ (:119-155).

Reference: per cloud, on the host -- concatenate parts, tile small clouds, np.random.permutation, one fancy-index per
array, scale by norm_factor, build the masks; then np.stack over the batch.  Here the whole ragged batch is packed once
(one 18-channel row per raw point) and ONE kernel launch (ancsh_input_sample) gathers / scales / one-hots every cloud
straight into the (B, N, .) tensors the network and the test-time losses read.  The permutation is an explicit input
(replay numpy's stream for parity) or drawn on the device.
"""
import numpy as np
import torch

from . import _lib

# channel layout of a packed raw row
_COLS = (("parts_pts", 3), ("parts_cls", 1), ("nocs_p", 3), ("nocs_g", 3), ("offset_heatmap", 1), ("offset_unitvec", 3),
         ("joint_orient", 3), ("joint_cls", 1))
NCHAN = sum(c for _, c in _COLS)
_CLS_COL, _JCLS_COL = 3, NCHAN - 1
# (record key, first output channel, width) in the (B, N, NCHAN-3) gathered tensor
_OUT = (("cls_gt", 0, 1), ("nocs_gt", 1, 3), ("nocs_gt_g", 4, 3), ("heatmap_gt", 7, 1), ("unitvec_gt", 8, 3), ("orient_gt", 11, 3),
        ("joint_cls_gt", 14, 1))


def pack_cloud(parts):
    """parts: dict keyed like create_data_shape2motion's return lists (parts_pts, parts_cls, nocs_p, nocs_g, offset_heatmap,
    offset_unitvec, joint_orient, joint_cls), each a list of per-part arrays or one concatenated array -> (n_raw, 18) float32."""
    cols = []
    for key, width in _COLS:
        v = parts[key]
        v = np.concatenate(v, axis=0) if isinstance(v, (list, tuple)) else np.asarray(v)
        cols.append(np.asarray(v, np.float32).reshape(v.shape[0], width))
    return np.ascontiguousarray(np.concatenate(cols, axis=1))


def tiled_size(n_raw, num_points):
    """Rows of the cloud after the reference's tiling rule (lib/dataset.py:290-293)."""
    return n_raw if n_raw >= num_points else (int(num_points / n_raw) + 1) * n_raw


def create_unit_data_batch(clouds, num_points, norm_factors, n_parts, perms=None, seed=None, device="cuda:0"):
    """clouds: list of `parts` dicts (see pack_cloud) or packed (n_raw, 18) arrays; norm_factors: one float per cloud
    (norm_factors[0] of the instance, lib/dataset.py:346); perms: optional list of int permutations of the TILED clouds
    (np.random.permutation(tiled_size)), else drawn on the device from `seed`.
    Returns the nocs_type 'A' record (:378-391) as (B, N, .) float32 device tensors:
    P, cls_gt, mask_array, nocs_gt, nocs_gt_g, heatmap_gt, unitvec_gt, orient_gt, joint_cls_gt, joint_cls_mask."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("articulated-pose_amd ops run on the MI355X only (no CPU fallback in the product path)")
    packed = [c if isinstance(c, np.ndarray) else pack_cloud(c) for c in clouds]
    B = len(packed)
    if B == 0:
        raise ValueError("create_unit_data_batch: empty batch")
    if any(p.ndim != 2 or p.shape[1] != NCHAN or p.shape[0] == 0 for p in packed):
        raise ValueError("every cloud must pack to a non-empty (n_raw, %d) array" % NCHAN)
    sizes = np.asarray([p.shape[0] for p in packed], np.int64)
    offsets = np.zeros(B + 1, np.int32)
    offsets[1:] = np.cumsum(sizes)
    rows = torch.from_numpy(np.concatenate(packed, axis=0)).to(dev)
    if perms is not None:
        if len(perms) != B or any(len(p) < num_points for p in perms):
            raise ValueError("perms: one permutation of the tiled cloud (>= num_points entries) per cloud")
        # numpy's fancy index (lib/dataset.py:298-300) takes an entry in [-size, -1] as size + entry and raises IndexError outside
        # [-size, size); the kernel reads row perm % n_raw of the raw cloud, so entries are normalised / refused here
        host = []
        for p, n in zip(perms, sizes):
            p = np.asarray(p[:num_points]).astype(np.int64)
            size = tiled_size(int(n), num_points)
            if p.size and (int(p.min()) < -size or int(p.max()) >= size):
                bad = int(p.min()) if int(p.min()) < -size else int(p.max())
                raise IndexError("perms: index %d is out of bounds for the tiled cloud of %d rows" % (bad, size))
            host.append(np.where(p < 0, p + size, p))
        perm = torch.from_numpy(np.stack([p.astype(np.int32) for p in host])).to(dev)
    else:
        g = torch.Generator(device=dev)
        g.manual_seed(0 if seed is None else int(seed))
        perm = torch.stack([torch.randperm(tiled_size(int(n), num_points), generator=g, device=dev)[:num_points].to(torch.int32)
                            for n in sizes])
    nf = torch.tensor(np.asarray(norm_factors, np.float32).reshape(B), device=dev)
    f = dict(dtype=torch.float32, device=dev)
    P = torch.empty((B, num_points, 3), **f)
    chan = torch.empty((B, num_points, NCHAN - 3), **f)
    mask_array = torch.empty((B, num_points, n_parts), **f)
    joint_cls_mask = torch.empty((B, num_points), **f)
    off = torch.from_numpy(offsets).to(dev)
    _lib.call("ancsh_input_sample", B, int(num_points), NCHAN, _lib.ptr(rows), _lib.ptr(off), _lib.ptr(perm), _lib.ptr(nf),
              _CLS_COL, _JCLS_COL, int(n_parts), _lib.ptr(P), _lib.ptr(chan), _lib.ptr(mask_array), _lib.ptr(joint_cls_mask))
    out = {"P": P, "mask_array": mask_array, "joint_cls_mask": joint_cls_mask}
    for key, c0, w in _OUT:
        out[key] = chan[:, :, c0] if w == 1 else chan[:, :, c0:c0 + w]
    return out
