"""Multi-GPU layout of the hot path: clouds are independent units through both halves (inference-mode
BN, per-cloud pose fit), so a batch is SHARDED across one-process-per-GPU ranks with no data-path
collective, and ONE gather (RCCL over xGMI; backend "nccl" on ROCm) of fixed-size per-cloud pose
records closes the batch.  The reference has no distributed code at all: its only parallelism is
multiprocessing.Process over contiguous slices (evaluation/pose_multi_process.py:53-67), whose
partition rule is kept here."""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist

CHILD_ENV = "ANCSH_LOCAL_RANK_CHILD"      # set in the ranks launch_local_ranks() starts: they must not launch again


ADDR_IN_USE_STATUS = 98                  # exit status of a rank whose rendezvous port was taken (errno EADDRINUSE): the launcher retries


def free_port():
    """A port that was free a moment ago.  Closing the probe socket and binding it again in rank 0 is a race; a rank that loses it
    exits with ADDR_IN_USE_STATUS (init_process_group below) and launch_local_ranks starts over on another port."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_process_group(backend, **kw):
    """dist.init_process_group for a rank started by launch_local_ranks: a rendezvous port that is already bound ends the rank
    with ADDR_IN_USE_STATUS instead of a traceback, which tells the launcher to retry on a fresh port."""
    try:
        dist.init_process_group(backend, **kw)
    except Exception as e:      # torch raises DistNetworkError / RuntimeError depending on where the bind fails
        msg = str(e)
        if os.environ.get(CHILD_ENV) == "1" and ("EADDRINUSE" in msg or "address already in use" in msg.lower()):
            print("rank %s: rendezvous port %s is in use" % (os.environ.get("RANK"), os.environ.get("MASTER_PORT")), file=sys.stderr, flush=True)
            sys.exit(ADDR_IN_USE_STATUS)
        raise


def init_groups(backend, device=None, probe_timeout_s=180.0):
    """Process groups of one rank: the CONTROL plane (barrier, max over ranks, rank identities) always on gloo, the DATA plane
    (the record gather) on RCCL (`backend` "nccl") when RCCL works on EVERY rank -- decided together: each rank creates the RCCL
    group and runs one probe all_reduce on `device`, the ranks then agree over gloo (MIN of their flags), so either all of them
    gather over RCCL or all of them stage the gather through host memory.  A node whose RCCL cannot start (no usable IPC between
    the processes, two ranks on one GPU, ...) therefore still completes the job and SAYS so.
    Returns (data_group or None, note): None = gather on the default gloo group (RecordGatherer stages through the host)."""
    init_process_group("gloo")
    world = dist.get_world_size()
    if backend != "nccl":
        return None, "gloo (host-staged), as requested"
    import datetime
    ok, err, g = 1, "", None
    try:
        kw = dict(backend="nccl", timeout=datetime.timedelta(seconds=probe_timeout_s))
        try:
            g = dist.new_group(device_id=torch.device(device), **kw)
        except TypeError:                      # a torch whose new_group has no device_id
            g = dist.new_group(**kw)
        t = torch.ones(1, device=device)
        dist.all_reduce(t, group=g)
        if int(t.item()) != world:
            ok, err = 0, "probe all_reduce over %d ranks returned %r" % (world, t.item())
    except Exception as e:                     # DistBackendError / RuntimeError: whatever the backend raises when it cannot start
        ok, err = 0, "%s: %s" % (type(e).__name__, " ".join(str(e).split())[:240])
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return g, "RCCL"
    errs = [None] * world
    dist.all_gather_object(errs, err)
    bad = ["rank %d: %s" % (r, e) for r, e in enumerate(errs) if e]
    return None, "gloo (host-staged): the RCCL probe failed (%s)" % "; ".join(bad[:2])


def rank_environment(rank, world, port, base=None):
    """Environment of local rank `rank` of `world`: what torch.distributed.run would export for one node."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC; a value the user set is kept
    env[CHILD_ENV] = "1"
    return env


def wants_self_launch(n_ranks, environ=None):
    """True when `--gpus n_ranks` was given to a plain `python script.py` (no launcher exported WORLD_SIZE > 1 and this
    process is not itself a rank started by launch_local_ranks)."""
    environ = os.environ if environ is None else environ
    return n_ranks > 1 and int(environ.get("WORLD_SIZE", "1")) == 1 and environ.get(CHILD_ENV) != "1"


def launch_local_ranks(n_ranks, argv, port=None, timeout=None, attempts=3):
    """Start `n_ranks` copies of `argv` (one process per GPU of this node), rank k with RANK = LOCAL_RANK = k, and wait for all
    of them: the counterpart of the reference's only parallel entry (evaluation/pose_multi_process.py:53-67 -- one Process per
    contiguous slice, start all, join all), with the torch.distributed environment so the ranks can form the RCCL group.
    Rank 0 inherits stdout (its JSON line / report is the launcher's output); the stdout of every other rank goes to the
    launcher's stderr, so a rank's own prints cannot interleave with rank 0's line.  Returns 0 when every rank exited 0; when one
    fails the others are terminated (by PID) and its exit code is returned.  With an automatically chosen port, a rank that found
    the port taken (ADDR_IN_USE_STATUS) makes the launcher start over on another port, `attempts` times at most."""
    auto = port is None
    rc = 0
    for _attempt in range(max(1, attempts)):
        rc = _launch_once(n_ranks, argv, free_port() if auto else port, timeout)
        if not (auto and rc == ADDR_IN_USE_STATUS):
            break
    return rc


def _launch_once(n_ranks, argv, port, timeout):
    procs = [subprocess.Popen(list(argv), env=rank_environment(k, n_ranks, port), stdout=None if k == 0 else sys.stderr)
             for k in range(n_ranks)]
    rc = 0
    try:
        import time
        t0 = time.time()
        alive = set(range(n_ranks))
        while alive:
            for k in sorted(alive):
                r = procs[k].poll()
                if r is None:
                    continue
                alive.discard(k)
                if r != 0 and rc == 0:
                    rc = r
                    print("rank %d exited with status %d: stopping the other ranks" % (k, r), file=sys.stderr, flush=True)
                    for j in alive:
                        procs[j].terminate()
            if timeout is not None and time.time() - t0 > timeout and alive:
                rc = rc or 124
                for j in alive:
                    procs[j].terminate()
                timeout = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def rank_identity(device=None):
    """Who this rank is: {rank, local_rank, pid, device_index, device_name, device_uuid, pci_bus_id} (device fields None on CPU)."""
    info = dict(rank=int(os.environ.get("RANK", "0")), local_rank=int(os.environ.get("LOCAL_RANK", "0")), pid=os.getpid(),
                device_index=None, device_name=None, device_uuid=None, pci_bus_id=None)
    if device is not None and torch.device(device).type == "cuda":
        idx = torch.device(device).index
        idx = torch.cuda.current_device() if idx is None else idx
        pr = torch.cuda.get_device_properties(idx)
        info.update(device_index=idx, device_name=pr.name, device_uuid=str(getattr(pr, "uuid", "")) or None)
        bus = [getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        if bus[1] is not None:
            info["pci_bus_id"] = "%04x:%02x:%02x" % (bus[0] or 0, bus[1], bus[2] or 0)
    return info


def all_rank_identities(device=None, group=None):
    """[rank_identity of rank 0, 1, ...] on every rank (one all_gather_object): the proof of who took part in a run."""
    me = rank_identity(device)
    if not (dist.is_available() and dist.is_initialized()):
        return [me]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, me, group=group)
    return out


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
        self.rank = dist.get_rank(group)            # group-local
        # `dst` is a GLOBAL rank (what dist.gather takes); compare it in the group's own numbering
        self.dst_local = dst if group is None else dist.get_group_rank(group, dst)
        self.host_staged = dist.get_backend(group) != "nccl"
        self.device = torch.device("cpu") if self.host_staged else torch.device(device)
        self._lanes = {}

    def buffers(self, lane=0):
        """dst: the `world` receive buffers of a lane (valid once that lane's gather has completed); None elsewhere."""
        if self.rank != self.dst_local:
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
    return torch.cat(parts, dim=0).to(local.device)      # a host-staged (gloo) gather of device records goes back to their device
