"""CPU test of the N>1 path (gloo, world_size 2): contiguous sharding with the reference's partition
rule and the single end-of-batch gather of pose records reassemble the global order."""
import datetime
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_matches_reference_rule():
    from articulated_pose_amd.dist import shard_range
    for n in (0, 1, 7, 32, 33, 100):
        for w in (1, 2, 4, 8):
            cover = []
            num_per = int(n / w) + 1                       # pose_multi_process.py:55
            for k in range(w):
                s, e = shard_range(n, w, k)
                assert s == min(num_per * k, n) and e == min(num_per * (k + 1), n)   # :61
                cover += list(range(s, e))
            assert cover == list(range(n))


def _run_ranks(target, extra=(), world=2, attempts=3):
    """Start `world` spawned ranks of `target(rank, world, port, *extra, q)` on a rendezvous port that was free a moment ago (NOT a fixed
    number: pid arithmetic landed inside the kernel's ephemeral range, where any outgoing connection may sit on it -- one rank then fails
    to bind and the other retries its connect for half an hour), wait for rank 0's result, and never leave a rank behind: a rank still
    alive after the wait is killed by PID, so that a failure here fails this test instead of hanging pytest at exit.  A lost port race
    (rank 0 exits before it reports) starts over on another port."""
    from articulated_pose_amd.dist import free_port
    import queue
    ctx = mp.get_context("spawn")
    for attempt in range(attempts):
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=target, args=(r, world, port) + tuple(extra) + (q,), daemon=True) for r in range(world)]
        for p in procs:
            p.start()
        got = None
        try:
            got = q.get(timeout=180)
            for p in procs:
                p.join(timeout=180)
        except queue.Empty:
            pass
        finally:
            for p in procs:
                if p.is_alive():
                    p.kill()
                    p.join(timeout=30)
        if got is not None:
            assert [p.exitcode for p in procs] == [0] * world
            return got
    raise AssertionError("no result from rank 0 in %d attempts (exit codes %s)" % (attempts, [p.exitcode for p in procs]))


def _worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import articulated_pose_amd  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from articulated_pose_amd.dist import gather_records, shard_range
    s, e = shard_range(n_total, world, rank)
    # record of cloud i = (K=3, 26) doubles filled with i -- stands for [baseline | nonlinear] models
    local = torch.arange(s, e, dtype=torch.float64).view(-1, 1, 1).expand(e - s, 3, 26).contiguous()
    out = gather_records(local, n_total, dst=0)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [9, 32])
def test_gather_records_world2_gloo(n_total):
    got = _run_ranks(_worker, (n_total,))
    assert got.shape == (n_total, 3, 26)
    np.testing.assert_array_equal(got[:, 0, 0], np.arange(n_total))


def _lanes_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import articulated_pose_amd  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from articulated_pose_amd.dist import RecordGatherer
    B, K = 4, 3
    g = RecordGatherer((B, K, 26), torch.float64, "cpu", dst=0)
    # three batches in flight (lanes), two rounds each: what bench.py does per step with world > 1
    for rnd in range(2):
        for lane in ("a", "b", "c"):
            val = 1000 * rnd + 100 * (ord(lane) - 97) + 10 * rank
            local = (torch.arange(B, dtype=torch.float64) + val).view(B, 1, 1).expand(B, K, 26).contiguous()
            g.gather(local, lane=lane)
    try:
        g.gather(torch.zeros((B + 1, K, 26), dtype=torch.float64))
        ok = False
    except ValueError:
        ok = True
    if rank == 0:
        q.put((ok, {lane: g.assembled(lane)[:, 0, 0].numpy() for lane in ("a", "b", "c")}))
    else:
        assert g.buffers("a") is None and ok
    dist.barrier()
    dist.destroy_process_group()


def test_record_gatherer_lanes_world2_gloo():
    """bench.py's per-step collective: fixed-size records, one set of receive buffers per batch in flight."""
    ok, got = _run_ranks(_lanes_worker)
    assert ok
    for i, lane in enumerate("abc"):
        want = np.concatenate([np.arange(4) + 1000 + 100 * i, np.arange(4) + 1000 + 100 * i + 10])   # rank order = global order
        np.testing.assert_array_equal(got[lane], want)


def test_self_launch_decision_and_environment():
    """`python bench.py --gpus N` without a launcher becomes the launcher; a rank it started (or one torchrun started) does not."""
    from articulated_pose_amd import dist as D
    assert not D.wants_self_launch(1, {})
    assert D.wants_self_launch(2, {}) and D.wants_self_launch(8, {"WORLD_SIZE": "1"})
    assert not D.wants_self_launch(2, {"WORLD_SIZE": "2", "RANK": "1"})                  # under torch.distributed.run
    env = D.rank_environment(3, 8, 23456, base={"PATH": "/bin"})
    assert env["RANK"] == env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["MASTER_ADDR"] == "127.0.0.1"
    assert env["MASTER_PORT"] == "23456" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    assert not D.wants_self_launch(8, env)                                                # the ranks do not launch again


_RANK_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import articulated_pose_amd
import torch.distributed as dist
from articulated_pose_amd import dist as D
if D.wants_self_launch(int(sys.argv[2])):
    sys.exit(D.launch_local_ranks(int(sys.argv[2]), [sys.executable] + sys.argv))
dist.init_process_group("gloo")
ids = D.all_rank_identities("cpu")
if len(sys.argv) > 3 and dist.get_rank() == 1:
    os._exit(7)          # dies without interpreter teardown (sys.exit with a live gloo group can end in SIGABRT instead of the status)
if dist.get_rank() == 0:
    print(json.dumps(ids), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_launch_local_ranks_world2_gloo(tmp_path):
    """The self-launch path end to end on CPU: one plain `python script --gpus 2` -> two ranks with a gloo group, rank 0 reports
    who took part (all_rank_identities); a failing rank's status comes back and the launcher does not hang."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ranks.py"
    script.write_text(_RANK_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(script), root, "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ids = json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])
    assert [i["rank"] for i in ids] == [0, 1] and [i["local_rank"] for i in ids] == [0, 1]
    assert ids[0]["pid"] != ids[1]["pid"] and ids[0]["device_index"] is None
    r = subprocess.run([sys.executable, str(script), root, "2", "fail"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])


_RANK8_SCRIPT = r'''
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import articulated_pose_amd
import torch
import torch.distributed as dist
from articulated_pose_amd import dist as D
world, mode = int(sys.argv[2]), sys.argv[3]
if D.wants_self_launch(world):
    sys.exit(D.launch_local_ranks(world, [sys.executable] + sys.argv, timeout=240))
print("rank %s says hello on its own stdout" % os.environ["RANK"], flush=True)      # ranks > 0: must not reach the launcher's stdout
D.init_process_group("gloo")
rank = dist.get_rank()
ids = D.all_rank_identities("cpu")
out = {}
for n_total in (128, 130, 5):                 # 128 = configs[4]'s batch; 130 and 5 leave ragged / empty last shards
    s, e = D.shard_range(n_total, world, rank)
    local = torch.arange(s, e, dtype=torch.float64).view(-1, 1, 1).expand(e - s, 3, 26).contiguous()
    got = D.gather_records(local, n_total, dst=0)
    if rank == 0:
        out[str(n_total)] = got[:, 0, 0].tolist()
        assert tuple(got.shape) == (n_total, 3, 26)
if mode == "fail" and rank == 5:
    os._exit(9)                               # one rank dies mid-run (no interpreter teardown: with a live gloo group sys.exit can end in
                                              # std::terminate -> SIGABRT instead of the status); the others are blocked in the barrier below
if rank == 0:
    print(json.dumps({"ids": ids, "gathered": out}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_eight_rank_preflight_gloo(tmp_path):
    """The 8-GPU launch of configs[4] without an 8-GPU node: `launch_local_ranks(8, ...)` with gloo on CPU -- eight ranks form the
    group and are listed, the ragged contiguous shards of 128 / 130 / 5 records come back in global order through the one gather,
    only rank 0 owns the launcher's stdout, and a rank that dies mid-run brings its status back instead of a hang
    (evaluation/pose_multi_process.py:53-67 is the reference's counterpart)."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ranks8.py"
    script.write_text(_RANK8_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, str(script), root, "8", "ok"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert [i["rank"] for i in res["ids"]] == list(range(8)) and len({i["pid"] for i in res["ids"]}) == 8
    for n_total in (128, 130, 5):
        assert res["gathered"][str(n_total)] == [float(i) for i in range(n_total)]
    hello = [l for l in r.stdout.splitlines() if "says hello" in l]
    assert hello == ["rank 0 says hello on its own stdout"]                       # the other ranks' stdout went to stderr
    assert sum("says hello" in l for l in r.stderr.splitlines()) == 7
    t0 = time.time()
    r = subprocess.run([sys.executable, str(script), root, "8", "fail"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 9, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 200 and "rank 5 exited with status 9" in r.stderr


def test_launcher_retries_when_the_port_was_taken(tmp_path, monkeypatch):
    """free_port() closes its probe socket before rank 0 binds the port: a rank that loses that race exits with
    ADDR_IN_USE_STATUS and the launcher starts over on another port (a caller-chosen port is not retried)."""
    import sys
    from articulated_pose_amd import dist as D
    marker = tmp_path / "seen"
    script = tmp_path / "r.py"
    script.write_text("import os, sys\np = os.environ['MASTER_PORT']\nopen(%r, 'a').write(p + '\\n')\n"
                      "sys.exit(%d if len(open(%r).read().split()) < 3 else 0)\n" % (str(marker), D.ADDR_IN_USE_STATUS, str(marker)))
    assert D.launch_local_ranks(1, [sys.executable, str(script)]) == 0
    assert len(marker.read_text().split()) == 3                                   # two refused attempts, then success
    marker.write_text("")
    assert D.launch_local_ranks(1, [sys.executable, str(script)], port=D.free_port()) == D.ADDR_IN_USE_STATUS
    env = D.rank_environment(0, 1, 1, base={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"                               # a value the user set is kept


_GROUPS_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import articulated_pose_amd
import torch
import torch.distributed as dist
from articulated_pose_amd import dist as D
if D.wants_self_launch(2):
    sys.exit(D.launch_local_ranks(2, [sys.executable] + sys.argv, timeout=240))
group, note = D.init_groups(sys.argv[2].split("+")[0], "cpu", probe_timeout_s=60, precheck=not sys.argv[2].endswith("+probe"))
rank = dist.get_rank()
g = D.RecordGatherer((4, 3, 26), torch.float64, "cpu", dst=0, group=group)
bufs = g.gather(torch.full((4, 3, 26), float(rank), dtype=torch.float64))
if rank == 0:
    print(json.dumps({"note": note, "group_is_none": group is None, "host_staged": g.host_staged,
                      "got": [float(b[0, 0, 0]) for b in bufs]}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_init_groups_agrees_on_the_fallback(tmp_path):
    """dist.init_groups with two CPU ranks: "gloo" is taken as requested; "nccl" cannot start here (no GPU), BOTH ranks see their probe
    fail, agree over the gloo control group and gather through the host -- the job completes and the note names the reason."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "groups.py"
    script.write_text(_GROUPS_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    # "nccl": the ranks learn from each other's identities that none has a GPU and never call RCCL; "nccl+probe" skips that check, so
    # BOTH probes fail with an exception and the MIN agreement over gloo is what is exercised
    for backend, want in (("gloo", "as requested"), ("nccl", "RCCL not attempted (rank(s) [0, 1] have no GPU)"),
                          ("nccl+probe", "the RCCL probe failed (rank 0: ")):
        r = subprocess.run([sys.executable, str(script), root, backend], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["group_is_none"] and out["host_staged"] and out["got"] == [0.0, 1.0]
        assert out["note"].startswith("gloo (host-staged)") and want in out["note"], out["note"]


def test_balanced_range_and_rccl_preconditions():
    from articulated_pose_amd.dist import balanced_range, rccl_preconditions
    assert [balanced_range(64, 4, r) for r in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]          # not 17,17,17,13
    assert [balanced_range(128, 8, r)[1] - balanced_range(128, 8, r)[0] for r in range(8)] == [16] * 8
    for n in (1, 7, 33, 130):
        for w in (1, 2, 4, 8):
            cover, sizes = [], []
            for r in range(w):
                s, e = balanced_range(n, w, r)
                cover += list(range(s, e))
                sizes.append(e - s)
            assert cover == list(range(n)) and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    ident = lambda r, idx, bus: dict(rank=r, device_index=idx, pci_bus_id=bus, device_uuid=None)
    assert rccl_preconditions([ident(0, 0, "0000:05:00"), ident(1, 1, "0000:15:00")]) == ""
    assert "share GPU 0000:05:00" in rccl_preconditions([ident(0, 0, "0000:05:00"), ident(1, 0, "0000:05:00")])
    assert rccl_preconditions([ident(0, 0, "0000:05:00"), ident(1, 0, "0000:05:00")], allow_shared=True) == ""
    assert "have no GPU" in rccl_preconditions([ident(0, 0, "0000:05:00"), ident(1, None, None)])
    assert "share GPU index 0" in rccl_preconditions([ident(0, 0, None), ident(1, 0, None)])


class _FakeSlot(object):
    def __init__(self):
        self.stream, self.out, self.P = None, None, None


class _FakePipeline(object):
    """AncshPipeline's step()/slot interface on CPU: the 'fit' of cloud i in step t writes 1000 * t + P[i, 0, 0] into its record."""

    def __init__(self, K, wa, wn, n_local, N, device, slots=1):
        self.K, self.n_local, self.slots, self._next, self.t = K, n_local, [_FakeSlot() for _ in range(slots)], 0, 0

    def load_inputs(self, P, joint_cls, pred=None, slot=None):
        assert len(P) == self.n_local == len(joint_cls)
        for sl in self.slots:
            sl.P = torch.as_tensor(P)

    def next_slot(self):
        return self.slots[self._next]

    def step(self):
        sl = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        sl.out = {"record": (1000.0 * self.t + sl.P[:, 0, 0].double()).view(-1, 1, 1).expand(self.n_local, self.K, 26).contiguous()}
        self.t += 1
        return sl, sl.out

    def synchronize(self):
        pass


def _sharded_worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import articulated_pose_amd  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from articulated_pose_amd.dist import ShardedPipeline, balanced_range
    sp = ShardedPipeline(3, None, None, n_total, 8, "cpu", slots=3, pipeline_factory=_FakePipeline)
    assert (sp.lo, sp.hi) == balanced_range(n_total, world, rank) and sp.lagged and sp.ragged == (n_total % world != 0)
    P = torch.arange(n_total, dtype=torch.float32).view(-1, 1, 1).expand(n_total, 8, 3).contiguous()
    sp.load_inputs(P, torch.zeros((n_total, 8), dtype=torch.int32))
    got = []
    for _ in range(7):                                   # more steps than slots: every slot is reused, its lagged gather happens first
        sp.step()
    sp.synchronize()
    last = sp.records()
    per_slot = [sp.records(sl) for sl in sp.pipe.slots]
    one = sp.solve(P, torch.zeros((n_total, 8), dtype=torch.int32))         # the one-call form: step 7
    if rank == 0:
        q.put((last[:, 0, 0].numpy(), [r[:, 0, 0].numpy() for r in per_slot], one[:, 0, 0].numpy(), tuple(one.shape)))
    else:
        assert last is None and one is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 9])
def test_sharded_pipeline_world2_gloo(n_total):
    """The product's multi-GPU entry (dist.ShardedPipeline) with two gloo ranks: balanced contiguous shards, one gather per batch per
    slot (host-staged, so lagged by one slot turn and drained by synchronize()), records on rank 0 in global cloud order -- for an
    even split and a ragged one."""
    last, per_slot, one, shape = _run_ranks(_sharded_worker, (n_total,))
    ids = np.arange(n_total, dtype=np.float64)
    np.testing.assert_array_equal(last, 6000 + ids)                         # step 6 = the most recent batch
    for k, rec in enumerate(per_slot):                                      # slots 0,1,2 hold steps 6,4,5
        np.testing.assert_array_equal(rec, 1000 * (6, 4, 5)[k] + ids)
    np.testing.assert_array_equal(one, 7000 + ids)
    assert shape == (n_total, 3, 26)


def test_sharded_pipeline_without_a_group_is_the_local_pipeline():
    from articulated_pose_amd.dist import ShardedPipeline
    sp = ShardedPipeline(2, None, None, 5, 4, "cpu", slots=2, pipeline_factory=_FakePipeline)
    assert (sp.world, sp.lo, sp.hi, sp.gatherer) == (1, 0, 5, None)
    P = torch.arange(5, dtype=torch.float32).view(-1, 1, 1).expand(5, 4, 3).contiguous()
    rec = sp.solve(P, torch.zeros((5, 4), dtype=torch.int32))
    assert tuple(rec.shape) == (5, 2, 26) and rec[:, 0, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
    with pytest.raises(ValueError):
        sp.load_inputs(P[:4], torch.zeros((4, 4), dtype=torch.int32))
