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


def init_groups(backend, device=None, probe_timeout_s=180.0, precheck=True):
    """Process groups of one rank: the CONTROL plane (barrier, max over ranks, rank identities) always on gloo, the DATA plane
    (the record gather) on RCCL (`backend` "nccl") when RCCL works on EVERY rank -- decided together: each rank creates the RCCL
    group and runs one probe all_reduce on `device`, the ranks then agree over gloo (MIN of their flags), so either all of them
    gather over RCCL or all of them stage the gather through host memory.  A node whose RCCL cannot start (no usable IPC between
    the processes, two ranks on one GPU, ...) therefore still completes the job and SAYS so.
    What can be known WITHOUT calling RCCL (a rank without a GPU, two ranks on one GPU: rccl_preconditions) is agreed first, so
    those cases never reach the probe.  Beyond that the agreement covers probes that FAIL WITH AN EXCEPTION on the ranks they fail
    on; a probe that fails on one rank by hanging leaves the healthy ranks inside new_group / all_reduce until `probe_timeout_s`,
    after which RCCL's watchdog may end those processes instead of raising: the job then dies (loudly) rather than falls back.
    Returns (data_group or None, note): None = gather on the default gloo group (RecordGatherer stages through the host)."""
    init_process_group("gloo")
    world = dist.get_world_size()
    if backend != "nccl":
        return None, "gloo (host-staged), as requested"
    import datetime
    # Cheap preconditions first, agreed over gloo BEFORE any rank touches RCCL: a probe that fails on one rank only would leave the
    # healthy ranks blocked inside new_group / all_reduce until RCCL's watchdog tears them down (no Python exception to catch), so
    # whatever can be known without RCCL is checked here, on every rank, from the same gathered list.
    why = rccl_preconditions(all_rank_identities(device), os.environ.get("ANCSH_SHARED_GPU_PROBE") == "1") if precheck else ""
    if why:
        return None, "gloo (host-staged): RCCL not attempted (%s)" % why
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


def rccl_preconditions(identities, allow_shared=False):
    """Why RCCL cannot carry the gather between these ranks, from their gathered rank_identity() records alone ('' = nothing known
    against it): a rank without a GPU, or two ranks on the same physical GPU (RCCL needs one per rank).  Every rank evaluates the
    same list, so all of them reach the same decision without a collective that could hang."""
    if any(i.get("device_index") is None for i in identities):
        return "rank(s) %s have no GPU" % [i["rank"] for i in identities if i.get("device_index") is None]
    if allow_shared:
        return ""
    seen = {}
    for i in identities:
        key = i.get("pci_bus_id") or i.get("device_uuid") or ("index", i.get("device_index"))
        if key in seen:
            return "ranks %d and %d share GPU %s" % (seen[key], i["rank"], key if isinstance(key, str) else "index %d" % key[1])
        seen[key] = i["rank"]
    return ""


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


def balanced_range(n_items, world_size, rank):
    """Contiguous slice of rank `rank` for an IN-MEMORY batch: shard sizes differ by at most one (64 over 4 -> 16,16,16,16; the
    reference's file-slicing rule shard_range() gives 17,17,17,13, and a step is as slow as its largest shard).  shard_range stays
    the rule wherever sub-pickle names depend on it (pose_multi_process.py)."""
    q, r = divmod(n_items, world_size)
    s = rank * q + min(rank, r)
    return s, s + q + (1 if rank < r else 0)


class _NoStream(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _on(stream):
    return _NoStream() if stream is None else torch.cuda.stream(stream)


class ShardedPipeline(object):
    """The north star's multi-GPU composition as ONE callable object: a global batch of depth clouds is split into balanced
    contiguous shards over one-process-per-GPU ranks, every rank runs its shard through its own AncshPipeline (both networks + the
    pose fit, no data-path collective), and ONE gather of the fixed-size (n, K, 26) f64 pose records on `dst` closes each batch --
    issued on the batch's own HIP stream behind the finish kernels (RCCL over xGMI), or staged through the host when the data
    group is gloo.  Counterpart of evaluation/pose_multi_process.py:53-67 (contiguous slices, one worker each, results joined).

    Without an initialised process group (or world 1) it is a plain AncshPipeline whose records() are the local ones.
    `pipeline_factory(num_parts, weights_ancsh, weights_npcs, n_local, num_points, device, slots=..., **kw)` builds the per-rank
    pipeline (default AncshPipeline; the gloo CPU tests inject a stand-in with the same step()/slot interface)."""

    def __init__(self, num_parts, weights_ancsh, weights_npcs, global_batch, num_points, device="cuda:0", data_group=None, dst=0,
                 slots=1, pipeline_factory=None, gather_single=False, **pipeline_kw):
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        self.rank = dist.get_rank() if self.distributed else 0
        if global_batch < self.world:
            raise ValueError("global_batch %d < %d ranks: a rank would hold no cloud" % (global_batch, self.world))
        self.K, self.global_batch, self.dst = num_parts, global_batch, dst
        self.lo, self.hi = balanced_range(global_batch, self.world, self.rank)
        self.n_local, self.n_max = self.hi - self.lo, -(-global_batch // self.world)
        self.ragged = global_batch % self.world != 0
        if pipeline_factory is None:
            from .pipeline import AncshPipeline as pipeline_factory
        self.pipe = pipeline_factory(num_parts, weights_ancsh, weights_npcs, self.n_local, num_points, device, slots=slots, **pipeline_kw)
        self.gatherer = RecordGatherer((self.n_max, num_parts, 26), torch.float64, device, dst, data_group) \
            if self.world > 1 or (gather_single and self.distributed) else None      # gather_single: run the collective even with one rank
        # a host-staged gather blocks the host on the batch it gathers: done for the batch just issued it would leave ONE batch in
        # flight, so a slot's record is gathered right before the slot is reused (its batch finished long ago); flush() drains
        self.lagged = self.gatherer is not None and self.gatherer.host_staged
        self._pending, self._pad = [], {}

    # ---- inputs -------------------------------------------------------------------------------------------------------------
    def load_inputs(self, P, joint_cls, pred=None, slot=None, is_global=True):
        """is_global: arrays hold the whole batch (global_batch leading) and this rank takes rows [lo, hi); else they ARE the shard."""
        if is_global:
            if len(P) != self.global_batch:
                raise ValueError("expected %d clouds, got %d" % (self.global_batch, len(P)))
            cut = lambda a: a[self.lo:self.hi]
            P, joint_cls = cut(P), cut(joint_cls)
            pred = None if pred is None else {k: cut(v) for k, v in pred.items()}
        self.pipe.load_inputs(P, joint_cls, pred, slot=slot)

    def load_draws(self, draws_a, draws_b, slot=None, is_global=True):
        if is_global:
            draws_a, draws_b = draws_a[self.lo:self.hi], None if draws_b is None else draws_b[self.lo:self.hi]
        self.pipe.load_draws(draws_a, draws_b, slot=slot)

    def prepare(self):
        self.pipe.prepare()
        return self

    # ---- one batch ----------------------------------------------------------------------------------------------------------
    def _gather(self, sl):
        rec = sl.out["record"]
        if self.ragged:                   # fixed-size collective: shards one cloud short are padded to the largest
            pad = self._pad.get(id(sl))
            if pad is None:
                pad = self._pad[id(sl)] = torch.zeros((self.n_max,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
            with _on(sl.stream):          # ordered behind the kernels that write the record (a host-staged gather synchronises this stream next)
                pad[: rec.shape[0]].copy_(rec)
            rec = pad
        self.gatherer.gather(rec, lane=id(sl), stream=sl.stream)

    def step(self):
        """Issue the next batch of this rank's shard (asynchronous) and its record gather.  Returns (slot, outputs)."""
        if self.lagged:
            nxt = self.pipe.next_slot()
            if nxt in self._pending:
                self._gather(nxt)
                self._pending.remove(nxt)
        sl, out = self.pipe.step()
        if self.gatherer is not None:
            if self.lagged:
                self._pending.append(sl)
            else:
                with _on(sl.stream):
                    self._gather(sl)
        return sl, out

    def flush(self):
        """Complete the gathers a host-staged data group still owes (no-op over RCCL: those are already enqueued)."""
        for sl in self._pending:
            self._gather(sl)
        self._pending = []

    def synchronize(self):
        self.flush()
        self.pipe.synchronize()

    def records(self, slot=None):
        """dst: the (global_batch, K, 26) f64 records of the batch last issued on `slot` (default: the most recent step), in global
        cloud order; None on the other ranks.  Valid after synchronize()."""
        sl = slot if slot is not None else self.pipe.slots[(self.pipe._next - 1) % len(self.pipe.slots)]
        if self.gatherer is None:
            return sl.out["record"]
        bufs = self.gatherer.buffers(id(sl))
        if bufs is None:
            return None
        parts = []
        for r in range(self.world):
            s, e = balanced_range(self.global_batch, self.world, r)
            parts.append(bufs[r][: e - s])
        return torch.cat(parts, dim=0)

    def solve(self, P, joint_cls, pred=None, is_global=True):
        """One global batch end to end: shard, run, gather, wait.  Returns records() (dst) / None (other ranks)."""
        self.load_inputs(P, joint_cls, pred, is_global=is_global)
        if getattr(self.pipe, "slots", None) and getattr(self.pipe.slots[0], "out", None) is None and hasattr(self.pipe, "prepare"):
            self.pipe.prepare()
        self.step()
        self.synchronize()
        return self.records()
