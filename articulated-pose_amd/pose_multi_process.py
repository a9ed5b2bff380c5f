"""Counterpart of `python evaluation/pose_multi_process.py --item=<cat> --domain=<seen|unseen> --nocs=ANCSH`
(evaluation/pose_multi_process.py:14-68).  The reference forks os.cpu_count()-2 worker processes over
contiguous slices of the test list; here every RANK (one process per MI355X) takes the slice the same rule
assigns it, fits its clouds in GPU batches, and writes the same per-worker
pickle  <base>/results/pickle/<exp>/subs/<baseline>_<domain>_<nocs>_<item>_rt_ours_0.1_<k>.pkl  (:60).
Like the reference, a plain `python -m articulated_pose_amd.pose_multi_process ...` starts its own workers (:53-67: one Process
per slice, start all, join all): one rank per visible GPU (`--gpus N` to choose); under `torchrun --nproc-per-node N` the
launcher's ranks are used as they are."""
import argparse
import os
import pickle
import sys
import time

from .dist import launch_local_ranks, shard_range, wants_self_launch
from .global_info import get_test_group, global_info
from .pose import solver_ransac_nonlinear


def _device_index():
    """LOCAL_RANK -> device; more ranks than GPUs (a test box) wrap around."""
    import torch
    return int(os.environ.get('LOCAL_RANK', 0)) % max(1, torch.cuda.device_count())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--domain', default='unseen', help='which sub test set to choose')
    ap.add_argument('--nocs', default='ANCSH', help='which sub test set to choose')
    ap.add_argument('--item', default='oven', help='object category for benchmarking')
    ap.add_argument('--base_path', default=None)
    ap.add_argument('--batch_size', type=int, default=32)
    ap.add_argument('--gpus', type=int, default=None, help='worker ranks to start, one per MI355X (default: every visible GPU)')
    args = ap.parse_args(argv)
    if 'WORLD_SIZE' not in os.environ or wants_self_launch(args.gpus or 0):
        import torch
        n_ranks = args.gpus or torch.cuda.device_count()
        if wants_self_launch(n_ranks):
            cmd = [sys.executable, '-m', 'articulated_pose_amd.pose_multi_process'] + list(sys.argv[1:] if argv is None else argv)
            raise SystemExit(launch_local_ranks(n_ranks, cmd))
    infos = global_info(args.base_path)
    d = infos.datasets[args.item]
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    my_dir = infos.base_path
    choose_threshold = 0.1
    all_test_h5 = os.listdir(os.path.join(my_dir, 'results/test_pred', d.exp))
    test_group = sorted(get_test_group(all_test_h5, d.test_list, domain=args.domain, spec_instances=d.spec_list))
    rt_file = os.path.join(my_dir, 'results/pickle', d.exp, '{}_{}_{}_rt.pkl'.format(args.domain, args.nocs, args.item))
    rts_all = pickle.load(open(rt_file, 'rb')) if os.path.exists(rt_file) else None     # GT poses are optional here
    directory = os.path.join(my_dir, 'results/pickle', d.exp, 'subs')
    os.makedirs(directory, exist_ok=True)
    s, e = shard_range(len(test_group), world, rank)
    sub = os.path.join(directory, '{}_{}_{}_{}_rt_ours_{}_{}.pkl'.format(d.baseline, args.domain, args.nocs, args.item, choose_threshold, rank))
    t0 = time.time()
    solver_ransac_nonlinear(s, e, d.exp, d.baseline, choose_threshold, d.num_parts, test_group, [], rts_all, sub,
                            base_path=my_dir, batch_size=args.batch_size, seed=rank,
                            device='cuda:%d' % _device_index())
    print('rank {}: {} clouds in {:.2f} s -> {}'.format(rank, e - s, time.time() - t0, sub))


if __name__ == '__main__':
    main()
