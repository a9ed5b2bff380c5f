"""Wire format between the network half and the pose-fit half: one record per cloud with the key
names of lib/prediction_io.py::save_batch_nn (:65-95).  HDF5 when h5py is importable (as the
reference), otherwise the same keys in an .npz next to it."""
import os

import numpy as np

try:
    import h5py
except ImportError:      # not installed in this image
    h5py = None

_GT_KEYS = (('cls_gt', 'cls_gt'), ('nocs_gt', 'nocs_gt'), ('nocs_gt_g', 'nocs_gt_g'), ('heatmap_gt', 'heatmap_gt'),
            ('unitvec_gt', 'unitvec_gt'), ('joint_axis_gt', 'orient_gt'), ('joint_cls_gt', 'joint_cls_gt'))


def _np(x):
    """Batch fields may be host arrays (record files) or device tensors (dataset.create_unit_data_batch)."""
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def record_for(pred_result, input_batch, b, is_mixed=False, W_reduced=True):
    instance_per_point = pred_result['W']
    if W_reduced:
        instance_per_point = np.argmax(instance_per_point, axis=2)
    rec = {
        'confidence_per_point': pred_result['confi_per_point'][b],
        'P': _np(input_batch['P'][b]),
        'nocs_per_point': pred_result['nocs_per_point'][b],
        'instance_per_point': instance_per_point[b],
        'heatmap_per_point': pred_result['heatmap_per_point'][b],
        'unitvec_per_point': pred_result['unitvec_per_point'][b],
        'joint_axis_per_point': pred_result['joint_axis_per_point'][b],
        'index_per_point': pred_result['index_per_point'][b],
    }
    if is_mixed:
        rec['gocs_per_point'] = pred_result['gocs_per_point'][b]
    for key, src in _GT_KEYS:
        if src in input_batch:
            rec[key] = _np(input_batch[src][b])
    return rec


def save_batch_nn(nn_name, pred_result, input_batch, basename_list, save_dir, sample_index=None, is_mixed=False,
                  W_reduced=True, two_stages=False):
    batch_size = pred_result['W'].shape[0]
    if batch_size != len(basename_list):
        raise ValueError('save_batch_nn: %d clouds in pred_result but %d basenames' % (batch_size, len(basename_list)))
    for b in range(batch_size):
        rec = record_for(pred_result, input_batch, b, is_mixed, W_reduced)
        if h5py is not None:
            with h5py.File(os.path.join(save_dir, basename_list[b] + '.h5'), 'w') as f:
                f.attrs['method_name'] = nn_name
                f.attrs['basename'] = basename_list[b]
                for k, v in rec.items():
                    f.create_dataset(k, data=v)
        else:
            np.savez(os.path.join(save_dir, basename_list[b] + '.npz'), method_name=nn_name,
                     basename=basename_list[b], **rec)


def load_record(save_dir, basename):
    """Read back one record as a dict of ndarrays (either container)."""
    p = os.path.join(save_dir, basename)
    if h5py is not None and os.path.exists(p + '.h5'):
        with h5py.File(p + '.h5', 'r') as f:
            return {k: f[k][()] for k in f.keys()}
    with np.load(p + '.npz', allow_pickle=False) as z:
        return {k: z[k] for k in z.files if k not in ('method_name', 'basename')}
