"""End-to-end hot path for a batch of depth point clouds on one MI355X:

    ANCSH forward  (joint axes / association heads)        main.py --test --nocs_type=ancsh
    NPCS  forward  (part-NOCS + part masks, the "baseline" network the pose stage reads when
                    USE_BASELINE, evaluation/parallel_ancsh_pose.py:232-237)   main.py --test --nocs_type=npcs
    pose fit       (per-part RANSAC + articulated LM)      evaluation/pose_multi_process.py

The whole step is ~100 asynchronous launches on one HIP stream; `AncshPipeline` captures them once
into a hipGraph and replays it per batch.
"""
import numpy as np
import torch

from .network import Network
from .pose import PoseSolver


def ensure_hardware_queues(slots, want=32):
    """Every batch in flight sits on its own HIP stream, and each stream needs its own hardware queue: with the runtime's default of 4
    a 1.6 ms stage-B kernel of one batch blocks other batches' kernels queued behind it (3.5 vs 2.35 ms/step measured at 4 vs 24
    queues).  The HIP runtime reads GPU_MAX_HW_QUEUES once, when it initialises.  Called by AncshPipeline for slots > 4: if the
    variable is unset and HIP is not up yet it is set here; if HIP is already initialised it cannot be changed any more -> a
    warning that names what the host should export.  Returns the effective value (None = unknown / runtime default)."""
    import os
    import warnings
    raw = os.environ.get("GPU_MAX_HW_QUEUES")
    try:
        cur = int(raw) if raw not in (None, "") else None
    except ValueError:
        warnings.warn("GPU_MAX_HW_QUEUES=%r is not an integer: ignored by this check (the HIP runtime decides what it makes of it)" % raw)
        cur = None
    if slots <= 4:
        return cur
    want = max(want, slots)                    # never fewer queues than batches in flight
    if cur is not None:
        if cur < slots:
            warnings.warn("AncshPipeline(slots=%d) with GPU_MAX_HW_QUEUES=%d: batches in flight will share hardware queues and wait behind "
                          "each other's long pose kernels; export GPU_MAX_HW_QUEUES>=%d before the process initialises HIP" % (slots, cur, slots))
        return cur
    if torch.cuda.is_initialized():
        warnings.warn("AncshPipeline(slots=%d): HIP is already initialised with the runtime's default of 4 hardware queues, so the %d batches "
                      "in flight will wait behind each other's long pose kernels (~1.5x slower steps).  Export GPU_MAX_HW_QUEUES=%d before the "
                      "first HIP call of the process (bench.py does so at its top)" % (slots, slots, want))
        return None
    os.environ["GPU_MAX_HW_QUEUES"] = str(want)
    return want


class _Slot(object):
    """Buffers + stream + captured graph of one batch in flight."""

    def __init__(self, B, N, K, device):
        f = dict(dtype=torch.float32, device=device)
        self.P = torch.zeros((B, N, 3), **f)
        self.joint_cls = torch.zeros((B, N), dtype=torch.int32, device=device)
        self.pred_nocs = torch.zeros((B, N, 3 * K), **f)
        self.pred_mask = torch.zeros((B, N, K), **f)
        self.pred_axis = torch.zeros((B, N, 3), **f)
        self.draws_a = self.draws_b = None      # optional replayed sample streams (see AncshPipeline.load_draws)
        self.stream = torch.cuda.Stream(device=device)
        self.graph = None
        self.out = None


class AncshPipeline(object):
    """step() runs the next batch through the whole path and returns that batch's outputs.

    couple=True : the pose stage reads the networks' own outputs (production data flow).
    couple=False: the pose stage reads `pred_*` buffers supplied by the caller -- used by the benchmark,
                  where random-init networks (no checkpoint ships with the reference) would hand the
                  fitter degenerate parts; every stage still runs inside the step.
    (Running one batch's own dependency graph on two streams -- [ANCSH net] || [NPCS net -> stage A], joined for stage B --
    was measured and dropped: both networks are matrix-pipe-bound even at 32 clouds, so their kernels time-slice instead of
    overlapping: 4.32 vs 4.45 ms for a lone batch, and 2.17 vs 1.75 ms/step with 16 batches in flight, where the 32 streams
    exceed the hardware queues.  Re-measured in round 3 with the geometry computed first and ONLY the NPCS network on the second
    stream: 17.2 k vs 20.1 k clouds/s at 16 batches in flight, 14.0 k vs 19.3 k at 8 -- two matrix-bound kernels interleaving
    their workgroups lose the XCD-local L2 reuse and each other's instruction-cache; coordinated batching beats concurrency.
    A narrower variant -- both networks' matrix-bound SA launches in order on the slot's stream, only the eleven small latency-bound
    launches of layer3 / fa_layer1 / fa_layer2 of the NPCS network forked to a side stream and joined before the tails (the two
    forwards driven phase by phase, outputs bit-identical) -- gained 1 % for a lone batch (8.58 k vs 8.48 k) and lost 15 % at 16
    batches in flight (17.2 k vs 20.1 k; 14.0 k vs 19.3 k at 8): every fork / join inside a batch costs more than it overlaps.  Likewise stage A || stage B of the fit on two streams -- they only share the partition --
    bought 0.13 ms of a lone batch's 4.29 ms and cost 0.44 ms/step at 16 batches in flight: dropped.)
    slots: batches kept in flight on separate HIP streams (round-robin).  The pose fit is latency-bound
           (a few hundred waves; a degenerate 3-point sample may run MINPACK's full 4200-evaluation budget in
           ONE lane, exactly as scipy does) while the networks are throughput-bound, so overlapping batch i's
           fit with batch i+1's networks keeps the CUs busy; results equal those of slots=1 (to the last bit for equal lm_schedule; slots <= 2 select the eight-lane LM schedule, see below)."""

    def __init__(self, num_parts, weights_ancsh, weights_npcs, batch_size, num_points, device="cuda:0",
                 inlier_th=0.1, niter_a=10000, niter_b=200, couple=True, use_graph=True, seed=0, slots=1, lm_schedule=None, tie_window=None,
                 arithmetic=None):
        self.K, self.B, self.N = num_parts, batch_size, num_points
        self.hw_queues = ensure_hardware_queues(max(1, slots))        # before the first HIP call this object makes
        self.device = torch.device(device)
        self.ancsh = Network(num_parts, weights_ancsh, "ancsh", device)
        self.npcs = Network(num_parts, weights_npcs, "npcs", device)
        # few batches in flight = a latency deployment: the LM fits take the eight-lanes-per-fit schedule (an EXPLICIT choice of this
        # class, overridable with lm_schedule; the C ABI's default schedule never depends on slots or batch size).  The two
        # schedules agree to ~1e-7, not to the last bit: pass lm_schedule="throughput" for bytes equal to a many-slot pipeline.
        # tie_window: None (default) = no tie statistics in the step (nobody reads them in a pipeline; the per-fit diagnostics of
        # PoseSolver -- tie_a / tie_b -- cost the stage-A finish kernel ~20 us a batch); pass parallel_ancsh_pose.TIE_WINDOW to get them
        self.solver = PoseSolver(num_parts, inlier_th, niter_a, niter_b, device,
                                 lm_schedule=lm_schedule or ("latency" if max(1, slots) <= 2 else "auto"), tie_window=tie_window)
        self.couple, self.seed = couple, seed
        # arithmetic of the shared-MLP layers: None = whatever ANCSH_SA_BF16X3 / ANCSH_SPLIT_SCHEME say (default: f32, the graded arithmetic);
        # "f32" | "bf16x3" | "f16x2" pins it for THIS pipeline (the split-16 experiment: bf16x3 at level 3, f16x2 at level 4 -- DESIGN section 8)
        if arithmetic not in (None, "f32", "bf16x3", "f16x2"):
            raise ValueError("arithmetic must be None, 'f32', 'bf16x3' or 'f16x2'")
        self.arithmetic = arithmetic
        # both networks layer by layer in grouped launches (paired.py; identical outputs); ANCSH_PAIRED=0: one forward after the other
        import os
        from .paired import PairedNetworks
        self.paired = PairedNetworks([self.ancsh, self.npcs]) if os.environ.get("ANCSH_PAIRED", "1") != "0" else None
        if self.paired is not None and not self.paired.eligible():
            self.paired = None
        self.slots = [_Slot(batch_size, num_points, num_parts, self.device) for _ in range(max(1, slots))]
        self._next = 0
        self._use_graph = use_graph
        self.stream = self.slots[0].stream

    # single-slot conveniences (slot 0)
    @property
    def P(self):
        return self.slots[0].P

    def load_inputs(self, P, joint_cls, pred=None, slot=None):
        for sl in (self.slots if slot is None else [self.slots[slot]]):
            sl.P.copy_(torch.as_tensor(P))
            sl.joint_cls.copy_(torch.as_tensor(np.asarray(joint_cls, np.int32)) if not torch.is_tensor(joint_cls) else joint_cls)
            if pred is not None:
                sl.pred_nocs.copy_(torch.as_tensor(pred["nocs_per_point"]))
                sl.pred_mask.copy_(torch.as_tensor(pred["instance_per_point"]))
                sl.pred_axis.copy_(torch.as_tensor(pred["joint_axis_per_point"]))

    def load_draws(self, draws_a, draws_b, slot=None):
        """Replay explicit 3-point sample streams (e.g. numpy's, `pose.parallel_ancsh_pose.draws_from_seed`) instead of the
        on-device generator: draws_a (B,K,niter_a,3), draws_b (B,K-1,niter_b,6) int32.  Call before prepare()."""
        for sl in (self.slots if slot is None else [self.slots[slot]]):
            sl.draws_a = torch.as_tensor(np.ascontiguousarray(draws_a, np.int32)).to(self.device)
            sl.draws_b = None if draws_b is None else torch.as_tensor(np.ascontiguousarray(draws_b, np.int32)).to(self.device)

    def _networks(self, P, geom):
        if self.paired is not None:
            return self.paired.predict(P, geom)          # every backbone layer of both networks in one grouped launch
        return self.ancsh.predict(P, geom), self.npcs.predict(P, geom)

    def _run(self, sl=None):
        sl = sl or self.slots[0]
        from . import pointnet_util
        geom = pointnet_util.Geometry()       # FPS / ball query / 3-NN depend only on P: computed once, used by both nets
        if self.arithmetic is None:
            a, n = self._networks(sl.P, geom)
        else:
            keep = pointnet_util.SA_BF16X3, pointnet_util.SPLIT_SCHEME
            level = {"f32": 0, "bf16x3": 3, "f16x2": 4}[self.arithmetic]
            pointnet_util.SA_BF16X3, pointnet_util.SPLIT_SCHEME = level, (self.arithmetic if level else keep[1])
            try:
                a, n = self._networks(sl.P, geom)
            finally:
                pointnet_util.SA_BF16X3, pointnet_util.SPLIT_SCHEME = keep
        if self.couple:
            nocs, mask, axis = n["nocs_per_point"], n["W"], a["joint_axis_per_point"]
        else:
            nocs, mask, axis = sl.pred_nocs, sl.pred_mask, sl.pred_axis
        sol = self.solver.solve(sl.P, nocs, mask, axis, sl.joint_cls, draws_a=sl.draws_a, draws_b=sl.draws_b, seed=self.seed)
        return dict(ancsh=a, npcs=n, pose=sol, record=sol["record"])      # (B, K, 26) float64, written by the fit's two finish kernels

    def prepare(self):
        torch.cuda.synchronize(self.device)
        for sl in self.slots:
            with torch.cuda.stream(sl.stream):
                for _ in range(2):
                    sl.out = self._run(sl)
            sl.stream.synchronize()
            if self._use_graph:
                sl.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(sl.graph, stream=sl.stream):
                    sl.out = self._run(sl)
        if self._use_graph:
            # first launches of the instantiated graphs (the runtime uploads an executable graph on its first launch), all slots in
            # flight together as in steady state; results are those of the eager passes above
            import os
            for _ in range(int(os.environ.get("ANCSH_PREPARE_REPLAYS", "2"))):
                for sl in self.slots:
                    with torch.cuda.stream(sl.stream):
                        sl.graph.replay()
            self.synchronize()
        return self

    def next_slot(self):
        """The slot the next step() will use: its outputs still hold the batch issued len(slots) steps ago (a consumer that must
        block the host for them -- e.g. a host-staged gather -- reads them here, when they have long been complete)."""
        return self.slots[self._next]

    def step(self):
        """Issue the next batch (asynchronous).  Returns (slot, outputs); outputs are valid once slot.stream is synchronised -- and
        only until the slot's NEXT step: a captured step owns its memory pool, so while a replay is in flight an output buffer may hold
        another tensor of the step (the pose record shares its block with the farthest-point indices, which the replay writes first).
        Whatever the caller enqueued on ITS current stream before calling step() (a clone or a gather of the slot's previous outputs) is
        ordered before the new batch: the slot's stream waits for that stream here.  A consumer on any other stream is the caller's to order."""
        sl = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        cur = torch.cuda.current_stream(self.device)
        if cur != sl.stream and not cur.query():     # an idle caller stream (the throughput loop) costs one query, no event and no barrier packet
            sl.stream.wait_stream(cur)
        with torch.cuda.stream(sl.stream):
            if sl.graph is not None:
                sl.graph.replay()
            else:
                sl.out = self._run(sl)
        return sl, sl.out

    def synchronize(self):
        for sl in self.slots:
            sl.stream.synchronize()
