#!/bin/bash
# First contact with a multi-GPU node.  No >= 2-GPU MI355X box has been available to any round, so the RCCL gather
# between two GPUs (the one collective of the path: evaluation/pose_multi_process.py:53-67 -> articulated_pose_amd.dist.RecordGatherer)
# has only ever run with gloo, several ranks sharing one GPU.  This script makes the first hardware run fail for hardware reasons only:
#   1. prints what the node shows (GPU count, xGMI topology) and refuses to go on with < 2 GPUs;
#   2. `bench.py --gpus 2 --steps 20 --only-timed` over RCCL with NCCL_DEBUG=INFO: the line must come back, name two ranks with DISTINCT
#      pci_bus_ids, and the transport lines of the log are printed (P2P/xGMI expected; SHM / NET means the node fell back);
#   3. the product's multi-GPU entry, dist.ShardedPipeline, on 2 ranks over RCCL against the single-process pipeline, bit for bit (the logic
#      of tests/test_dist_gpu.py::test_sharded_pipeline_equals_single_process with the nccl backend and one GPU per rank);
#   4. the drop-in entry point `python -m articulated_pose_amd.pose_multi_process` on 2 ranks over a synthetic results tree
#      (tests/test_entry_gpu.py's tree): both per-worker pickles written, their union = the single-rank pickle.
# Usage: tools/rccl_preflight.sh [N_GPUS=2]      (from the repo root; exits non-zero at the first failed check)
set -u
N=${1:-2}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0} MASTER_ADDR=127.0.0.1
OUT=${PREFLIGHT_OUT:-gpurun_out/rccl_preflight}
mkdir -p "$OUT"
fail() { echo "PREFLIGHT FAIL: $*"; exit 1; }

echo "== 1. node =="
NDEV=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "visible GPUs: $NDEV"
(rocm-smi --showtopo 2>/dev/null || true) | tee "$OUT/topo.txt" | head -40
[ "$NDEV" -ge "$N" ] || fail "needs $N GPUs, the node shows $NDEV (a 1-GPU box can only run the gloo stand-in: pytest tests/test_dist_gpu.py tests/test_bench_gpu.py)"

echo "== 2. bench.py --gpus $N over RCCL =="
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH timeout 900 python bench.py --gpus "$N" --steps 20 --warmup 5 --only-timed > "$OUT/bench.json" 2> "$OUT/bench.log" \
  || { tail -30 "$OUT/bench.log"; fail "bench.py --gpus $N did not finish (log: $OUT/bench.log)"; }
grep -E "NCCL INFO (Channel|Connected|.*via|.*comm .* rank)" "$OUT/bench.log" | head -20
python - "$OUT/bench.json" "$N" <<'PY' || fail "bench line"
import json, sys
line = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); n = int(sys.argv[2])
ranks = line["ranks"]
assert line["n_gpus"] == n and len(ranks) == n, (line["n_gpus"], len(ranks))
bus = [r["pci_bus_id"] for r in ranks]
assert len(set(bus)) == n, "ranks share a GPU: %s" % bus
assert "1 RCCL gather" in line["parallelism"], line["parallelism"]      # not the host-staged fallback
print("value %.0f clouds/s on %d GPUs (%.4f ms/step); ranks on %s" % (line["value"], n, line["ms_per_step"], bus))
PY

echo "== 3. dist.ShardedPipeline over RCCL == single-process pipeline =="
cat > "$OUT/shard.py" <<'PY'
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import articulated_pose_amd
from articulated_pose_amd import dist as D
from articulated_pose_amd.pose.parallel_ancsh_pose import draws_from_seed
from articulated_pose_amd.synthetic import make_cloud, make_predictions
from articulated_pose_amd.weights import synthetic_weights
world = int(sys.argv[1])
if D.wants_self_launch(world):
    sys.exit(D.launch_local_ranks(world, [sys.executable] + sys.argv))
rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
dev = torch.device("cuda", local % torch.cuda.device_count() if world > 1 else 0)
torch.cuda.set_device(dev)
group = None
if world > 1:
    import torch.distributed as dist
    group, note = D.init_groups("nccl", dev)
    assert group is not None and note == "RCCL", note          # the host-staged fallback is a FAIL here: this run is about RCCL
K, N, n_total, na, nb = 3, 512, 11, 200, 16                 # 11 clouds: a ragged split (6 + 5), padded fixed-size gather
clouds = [make_cloud(70 + i, N=N, K=K) for i in range(n_total)]
preds = [make_predictions(c, K, seed=i) for i, c in enumerate(clouds)]
da, db = [], []
for i, p in enumerate(preds):
    counts = np.bincount(np.argmax(p["instance_per_point"], 1), minlength=K)
    a, b = draws_from_seed(1000 + i, counts, na, nb)
    da.append(a); db.append(b)
st = lambda key, src: np.stack([x[key] for x in src])
sp = D.ShardedPipeline(K, synthetic_weights(K, seed=0), synthetic_weights(K, mixed_pred=False, early_split_nocs=False, seed=1), n_total, N, dev,
                       data_group=group, slots=2, niter_a=na, niter_b=nb, couple=False, use_graph=True, lm_schedule="throughput")
sp.load_draws(np.stack(da), np.stack(db))
rec = sp.solve(st("P", clouds), st("joint_cls_gt", preds), {k: st(k, preds) for k in ("nocs_per_point", "instance_per_point", "joint_axis_per_point")})
for _ in range(5):
    sp.step()                                                 # gathers of several batches in flight on their own slot streams
sp.synchronize()
if rank == 0:
    assert np.array_equal(sp.records().cpu().numpy(), rec.cpu().numpy(), equal_nan=True)
    np.save(sys.argv[2], rec.cpu().numpy())
if world > 1:
    dist.barrier(); dist.destroy_process_group()
PY
timeout 600 python "$OUT/shard.py" 1 "$OUT/rec1.npy" || fail "single-process fit"
NCCL_DEBUG=WARN timeout 600 python "$OUT/shard.py" "$N" "$OUT/recN.npy" || fail "sharded fit over RCCL"
python - "$OUT" <<'PY' || fail "gathered records differ from the single-process fit"
import sys, numpy as np
a, b = np.load(sys.argv[1] + "/rec1.npy"), np.load(sys.argv[1] + "/recN.npy")
assert a.shape == b.shape == (11, 3, 26) and np.array_equal(a, b, equal_nan=True), float(np.nanmax(np.abs(a - b)))
print("gathered records of the sharded fit == single-process records, bit for bit:", a.shape)
PY

echo "== 4. pose_multi_process on $N ranks =="
timeout 900 python -m pytest tests/test_entry_gpu.py -x -q -k "pose_multi_process or evaluation_sh" 2>&1 | tail -3
[ "${PIPESTATUS[0]}" -eq 0 ] || fail "entry-point tests"
echo "PREFLIGHT OK ($N GPUs)"
