"""Multi-GPU layout of the hot path: clouds are independent units through both halves (inference-mode
BN, per-cloud pose fit), so a batch is SHARDED across one-process-per-GPU ranks with no data-path
collective, and ONE gather (RCCL over xGMI; backend "nccl" on ROCm) of fixed-size per-cloud pose
records closes the batch.  The reference has no distributed code at all: its only parallelism is
multiprocessing.Process over contiguous slices (evaluation/pose_multi_process.py:53-67), whose
partition rule is kept here."""
import torch
import torch.distributed as dist


def shard_range(n_items, world_size, rank):
    """Contiguous slice of rank `rank`, the reference's rule (pose_multi_process.py:55,61):
    num_per = int(n/world) + 1; [num_per*k, min(num_per*(k+1), n))."""
    num_per = int(n_items / world_size) + 1
    s = min(num_per * rank, n_items)
    e = min(num_per * (rank + 1), n_items)
    return s, e


def gather_records(local, n_total, dst=0, group=None):
    """Gather per-cloud records (n_local, ...) of every rank on `dst` in global cloud order.
    Ranks hold shard_range() slices of n_total clouds (ragged: padded to the largest shard so that a
    single fixed-size gather suffices).  Returns the (n_total, ...) tensor on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    num_per = int(n_total / world) + 1
    pad = torch.zeros((num_per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        s, e = shard_range(n_total, world, r)
        parts.append(bufs[r][: e - s])
    return torch.cat(parts, dim=0)
