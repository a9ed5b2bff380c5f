"""Counterpart of `python evaluation/compute_gt_pose.py --item=<cat> --domain=<seen|unseen> --nocs=ANCSH --save`
(evaluation/compute_gt_pose.py:21-103): the ground-truth part poses every later evaluation step reads
(`rts_all` of pose_multi_process.py:45-51).

Reference: per record, per part, one numpy Umeyama (lib/aligning.py:580-622) between the GT part-NOCS and the camera
points, composed into a 4x4 by compose_rt (:14-19).  Here all (record, part) problems of a chunk go through ONE
ancsh_umeyama launch (pose.umeyama_batch) and the same pickle is written:
    <base>/results/pickle/<exp>/<domain>_<nocs>_<item>_rt.pkl = {basename: {'scale': {'gt': [s_j]}, 'rt': {'gt': [RT_j]}}}
"""
import argparse
import os
import pickle
import time

import numpy as np

from . import prediction_io
from .global_info import _RECORD_SUFFIXES, global_info
from .pose.aligning import umeyama_batch


def compose_rt(rotation, translation):
    """4x4 float32 [R^T | t] from Umeyama's returned (transposed) rotation and translation (compute_gt_pose.py:14-19)."""
    rt = np.eye(4, dtype=np.float32)
    rt[:3, :3] = np.asarray(rotation).T
    rt[:3, 3] = translation
    return rt


def get_full_test(all_test_h5, unseen_instances, domain='seen', spec_instances=[], category=None):
    """Every record of the held-out ('unseen') or of the other ('seen') instances -- no frame / articulation thinning --
    minus the special instances: the selection of lib/data_utils.py:936-957.  Input order is kept."""
    held_out, special = frozenset(unseen_instances), frozenset(spec_instances)
    keep_unseen = domain != 'seen'
    return [f for f in all_test_h5
            if f[:4] not in special and f.endswith(_RECORD_SUFFIXES) and ((f.split('.')[0].split('_')[0] in held_out) == keep_unseen)]


def gt_pose_records(records, num_parts, nocs='ANCSH', device='cuda:0'):
    """records: [(basename, dict with 'P', 'cls_gt', 'nocs_gt' (+ 'nocs_gt_g' for NAOCS))] -> {basename: rts_dict}.
    A part without points (or a record missing a field) is skipped like the reference's bare `except: pass` (:99-100)."""
    src, tgt, owner = [], [], []
    for basename, r in records:
        try:
            mask_gt = np.asarray(r['cls_gt'])
            nocs_gt = np.asarray(r['nocs_gt'] if nocs == 'ANCSH' else r['nocs_gt_g'])
            pts = np.asarray(r['P'])[:, :3]
            parts = [np.where(mask_gt == j)[0] for j in range(num_parts)]
            if any(len(p) == 0 for p in parts):
                continue                                 # the reference's SVD raises on an empty part -> record skipped
            for p in parts:
                src.append(nocs_gt[p, :])
                tgt.append(pts[p, :])
            owner.append(basename)
        except (KeyError, IndexError, ValueError):
            continue
    out = {}
    if not owner:
        return out
    fits = umeyama_batch(src, tgt, device)
    for i, basename in enumerate(owner):
        rt_gt, scale_gt = [], []
        for j in range(num_parts):
            s, r, t, _ = fits[i * num_parts + j]
            rt_gt.append(compose_rt(r, t))
            scale_gt.append(s)
        out[basename] = {'scale': {'gt': scale_gt}, 'rt': {'gt': rt_gt}}
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--domain', default='unseen', help='which sub test set to choose')
    ap.add_argument('--nocs', default='ANCSH', help='which nocs type to use')
    ap.add_argument('--item', default='oven', help='object category for benchmarking')
    ap.add_argument('--save', action='store_true', help='save err to pickles')
    ap.add_argument('--base_path', default=None)
    ap.add_argument('--chunk', type=int, default=256, help='records per Umeyama launch')
    args = ap.parse_args(argv)
    infos = global_info(args.base_path)
    d = infos.datasets[args.item]
    base = os.path.join(infos.base_path, 'results')
    pred_dir = os.path.join(base, 'test_pred', d.exp)
    test_group = get_full_test(sorted(os.listdir(pred_dir)), d.test_list, domain=args.domain, spec_instances=d.spec_list)
    print('we have {} testing data for {} {}'.format(len(test_group), args.domain, args.item))
    t0 = time.time()
    all_rts = {}
    dev = 'cuda:%d' % int(os.environ.get('LOCAL_RANK', 0))
    for c0 in range(0, len(test_group), args.chunk):
        names = [f.rsplit('.', 1)[0] for f in test_group[c0:c0 + args.chunk]]
        all_rts.update(gt_pose_records([(n, prediction_io.load_record(pred_dir, n)) for n in names], d.num_parts, args.nocs, dev))
    if args.save:
        out_dir = os.path.join(base, 'pickle', d.exp)
        os.makedirs(out_dir, exist_ok=True)
        file_name = os.path.join(out_dir, '{}_{}_{}_rt.pkl'.format(args.domain, args.nocs, args.item))
        with open(file_name, 'wb') as f:
            pickle.dump(all_rts, f, protocol=2)
        print('saving to ', file_name)
    print('{} GT poses in {:.2f} seconds'.format(len(all_rts), time.time() - t0))
    return all_rts


if __name__ == '__main__':
    main()
