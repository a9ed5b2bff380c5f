"""Counterpart of the reference's `python main.py --item=<cat> --nocs_type=<ancsh|npcs> --test`
(main.py:18-123 -> lib/network.py:257-316): run the network on every cloud of a test directory and
write one prediction record per cloud (lib/prediction_io.py:65-95 key names).

    python -m articulated_pose_amd.main --item eyeglasses --nocs_type ancsh --test \
        --weights weights.npz --data_dir <dir of *.npz|*.h5 with 'P' (+GT keys)> --out_dir results/test_pred/3.9

Weights: flat .npz keyed by the reference's TF variable names (weights.py); --weights synthetic:<seed>
builds seeded random weights (no pretrained checkpoint ships with the reference); --weights tf:<checkpoint prefix>
reads a TensorFlow-1 checkpoint directly (checkpoint.py, the reference's `model.ckpt-*` files: main.py:81-97)."""
import argparse
import os

import numpy as np

from . import prediction_io
from .global_info import global_info
from .network import Network
from .weights import load_npz, synthetic_weights


def iterate_batches(data_dir, batch_size, n_parts=None):
    names = sorted(f for f in os.listdir(data_dir) if f.endswith('.npz') or f.endswith('.h5'))
    for i in range(0, len(names), batch_size):
        chunk = [n.rsplit('.', 1)[0] for n in names[i:i + batch_size]]
        recs = [prediction_io.load_record(data_dir, n) for n in chunk]
        batch = {'basename_list': chunk, 'P': np.stack([r['P'][:, :3] for r in recs]).astype(np.float32)}
        for key, src in (('cls_gt', 'cls_gt'), ('nocs_gt', 'nocs_gt'), ('nocs_gt_g', 'nocs_gt_g'), ('heatmap_gt', 'heatmap_gt'),
                         ('unitvec_gt', 'unitvec_gt'), ('orient_gt', 'joint_axis_gt'), ('joint_cls_gt', 'joint_cls_gt'),
                         ('mask_array', 'mask_array'), ('joint_cls_mask', 'joint_cls_mask')):
            if all(src in r for r in recs):
                batch[key] = np.stack([r[src] for r in recs])
        # the two masks the test-time losses need are not part of a record (lib/prediction_io.py:73-92); the loader derives them
        # from the labels exactly as lib/dataset.py:353-357 does
        if n_parts and 'cls_gt' in batch and 'mask_array' not in batch:
            lab = batch['cls_gt'].astype(np.int8).astype(np.int64)
            mask = np.zeros(lab.shape + (n_parts,), np.float32)
            np.put_along_axis(mask, np.where(lab < 0, lab + n_parts, lab)[..., None], 1.0, axis=-1)
            batch['mask_array'] = mask
        if 'joint_cls_gt' in batch and 'joint_cls_mask' not in batch:
            batch['joint_cls_mask'] = (batch['joint_cls_gt'] > 0).astype(np.float32)
        yield batch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--item', default='eyeglasses')
    ap.add_argument('--nocs_type', default='ancsh', choices=['ancsh', 'npcs'])
    ap.add_argument('--test', action='store_true', help='Run network in test time')
    ap.add_argument('--gpu', default='0')
    ap.add_argument('--weights', default='synthetic:0')
    ap.add_argument('--data_dir', required=True)
    ap.add_argument('--out_dir', default=None)
    ap.add_argument('--batch_size', type=int, default=16)      # cfg/network_config.yml:12
    args = ap.parse_args(argv)
    if not args.test:
        raise SystemExit('inference-only build: pass --test (training is out of scope)')
    info = global_info().datasets[args.item]
    mixed = args.nocs_type == 'ancsh'
    if args.weights.startswith('synthetic:'):
        weights = synthetic_weights(info.num_parts, mixed_pred=mixed, early_split_nocs=mixed, seed=int(args.weights.split(':')[1]))
    elif args.weights.startswith('tf:'):
        from .checkpoint import is_model_variable, read_tf_checkpoint
        weights = read_tf_checkpoint(args.weights[3:], include=is_model_variable)
    else:
        weights = load_npz(args.weights)
    exp = info.exp if mixed else info.baseline                   # main.py:44,51
    out_dir = args.out_dir or os.path.join('results', 'test_pred', exp)
    net = Network(info.num_parts, weights, args.nocs_type, 'cuda:%s' % args.gpu.split(',')[0])
    res = net.predict_and_save(iterate_batches(args.data_dir, args.batch_size, info.num_parts), out_dir)
    print('wrote %d prediction records to %s' % (res['n'], out_dir))
    if res['msg']:
        print(res['msg'])


if __name__ == '__main__':
    main()
