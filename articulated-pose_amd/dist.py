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


class RecordGatherer(object):
    """The ONE collective of the hot path: a fixed-size gather of per-cloud result records on `dst`.

    Every rank contributes a (n_local, ...) tensor of identical shape; `dst` receives them in rank order, i.e. in global
    cloud order for contiguous shards.  Receive buffers are allocated once per `lane` (a batch in flight: gathers of
    different lanes may overlap), so the steady state allocates nothing.  With the "nccl" backend (= RCCL on ROCm) the
    gather is enqueued on the caller's current HIP stream behind the kernels that produce `local` -- no host
    synchronisation; with "gloo" (CPU tests, or several ranks sharing one GPU) the record is staged through host memory
    after synchronising `stream`."""

    def __init__(self, record_shape, dtype, device, dst=0, group=None):
        self.shape, self.dtype, self.dst, self.group = tuple(record_shape), dtype, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.host_staged = dist.get_backend(group) != "nccl"
        self.device = torch.device("cpu") if self.host_staged else torch.device(device)
        self._lanes = {}

    def buffers(self, lane=0):
        """dst: the `world` receive buffers of a lane (valid once that lane's gather has completed); None elsewhere."""
        if self.rank != self.dst:
            return None
        if lane not in self._lanes:
            self._lanes[lane] = [torch.empty(self.shape, dtype=self.dtype, device=self.device) for _ in range(self.world)]
        return self._lanes[lane]

    def gather(self, local, lane=0, stream=None):
        if tuple(local.shape) != self.shape or local.dtype != self.dtype:
            raise ValueError("record must be %s %s, got %s %s" % (self.shape, self.dtype, tuple(local.shape), local.dtype))
        if self.host_staged and local.is_cuda:
            if stream is not None:
                stream.synchronize()
            local = local.cpu()
        bufs = self.buffers(lane)
        dist.gather(local.contiguous(), bufs, dst=self.dst, group=self.group)
        return bufs

    def assembled(self, lane=0):
        """dst: (world * n_local, ...) records in global order."""
        bufs = self.buffers(lane)
        return None if bufs is None else torch.cat(bufs, dim=0)


def gather_records(local, n_total, dst=0, group=None):
    """Gather per-cloud records (n_local, ...) of every rank on `dst` in global cloud order.
    Ranks hold shard_range() slices of n_total clouds (ragged: padded to the largest shard so that a
    single fixed-size gather suffices).  Returns the (n_total, ...) tensor on dst, None elsewhere."""
    world = dist.get_world_size(group)
    num_per = int(n_total / world) + 1
    pad = torch.zeros((num_per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    g = RecordGatherer(pad.shape, pad.dtype, pad.device, dst, group)
    bufs = g.gather(pad)
    if bufs is None:
        return None
    parts = []
    for r in range(world):
        s, e = shard_range(n_total, world, r)
        parts.append(bufs[r][: e - s])
    return torch.cat(parts, dim=0)
