"""CPU test of the N>1 path (gloo, world_size 2): contiguous sharding with the reference's partition
rule and the single end-of-batch gather of pose records reassemble the global order."""
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


def _worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import articulated_pose_amd  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert got.shape == (n_total, 3, 26)
    np.testing.assert_array_equal(got[:, 0, 0], np.arange(n_total))


def _lanes_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import articulated_pose_amd  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_lanes_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert ok
    for i, lane in enumerate("abc"):
        want = np.concatenate([np.arange(4) + 1000 + 100 * i, np.arange(4) + 1000 + 100 * i + 10])   # rank order = global order
        np.testing.assert_array_equal(got[lane], want)
