#!/usr/bin/env python
"""Where one pipelined step goes: an occupancy-weighted account of a `rocprofv3 --kernel-trace` of the driver's command
(`python bench.py --only-timed`: warm-up + the timed steps, 16 batches in flight, nothing else).

Every dispatch contributes  duration x share,  share = min(1, waves / 1024)  (1024 SIMDs: a launch of >= 1024 waves can keep
every SIMD busy, a 64-wave LM launch occupies 1/16 of the chip however long it runs).  Two views per kernel family:

  busy_simd_ms    sum(duration x share) per step: the SIMD time the family asks for (can exceed the step when launches of
                  several batches overlap and time-share SIMDs)
  attributed_ms   a sweep over the trace's time line: in every interval the shares of the running dispatches are summed
                  (S); min(1, S) of the interval is occupied and is split between the running families in proportion to
                  their shares, 1 - min(1, S) is idle.  attributed + idle = the step, exactly.

  pipe_ms         (with --counters <sq_counters_per_kernel.csv>, tools/sq_counters.py) the same sweep with every running dispatch
                  weighted by the PIPE WORK it does per unit time instead of by the SIMDs it could sit on: a launch's work is its
                  kernel's SQ_VALU_MFMA_BUSY_CYCLES + 4 cycles x its non-MFMA SQ_INSTS_VALU per launch (measured with the kernel
                  alone on the chip), spread evenly over the launch's start-to-end time in THIS trace.  Equal SIMD shares
                  under-charge the matrix kernels (round 5: fused SA attributed 0.64 ms for 104.3 GFLOP = 163 TFLOP/s, above
                  the 157.3 peak); pipe_ms is the column to read as "whose instructions the SIMDs were issuing".
                  floor_ms = the family's pipe work / the whole chip's issue rate: what it would cost alone at 100 % issue.

usage: step_account.py <kernel_trace.csv> [--steps N] [--trim 0.15] [--counters sq.csv] [--out table.txt]
(--steps N: the timed region = the last N steps of the trace, found from a once-per-step marker kernel; --trim drops that fraction
of them at both ends: pipeline fill / drain)"""
import argparse
import collections
import csv
import math
import sys

FAMILIES = [
    ("sa1_fused", "fused SA (MFMA)"), ("sa2_fused", "fused SA (MFMA)"),
    ("mlp_chain", "tail chain (MFMA)"),
    ("conv1x1_few_rows", "fp partial product (VALU)"), ("fp_init_kernel", "fp partial product (VALU)"),
    ("sa3_chain", "conv1x1 family (MFMA)"), ("fp1_chain", "conv1x1 family (MFMA)"), ("fp2_chain", "conv1x1 family (MFMA)"),      # round 5: the mid-section chains
    ("conv_rowtile", "conv1x1 family (MFMA)"), ("conv_packed", "conv1x1 family (MFMA)"), ("conv1x1_kernel", "conv1x1 family (MFMA)"),
    ("conv_pair", "conv1x1 family (MFMA)"),
    ("fps_", "farthest point sampling"),
    ("query_ball", "ball query"), ("group_", "group / group_max"),
    ("three_nn", "3-NN + interpolate + concat"), ("three_weights", "3-NN + interpolate + concat"),
    ("three_interpolate", "3-NN + interpolate + concat"), ("fp_concat", "3-NN + interpolate + concat"), ("fp_interp", "3-NN + interpolate + concat"),
    ("head_act", "head activations"),
    ("partition", "pose: partition / median"), ("joint_direction", "pose: partition / median"),
    ("ransac_single_score", "pose stage A: scoring"), ("ransac_single_finish", "pose stage A: refit"), ("ransac_single", "pose stage A: other"),
    ("ransac_joint_lm", "pose stage B: LM fits"), ("ransac_joint_finish", "pose stage B: refit"), ("ransac_joint", "pose stage B: init / models"),
    ("umeyama", "pose: umeyama"),
]
SIMDS = 1024


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    if "ancsh::" in name:
        return "other ancsh"
    return "aten / runtime (copies, cat, fill)"


def col(row, *names):
    for n in names:
        if n in row and row[n] != "":
            return row[n]
    return None


def load(path):
    out = []
    for r in csv.DictReader(open(path)):
        if col(r, "Kind") not in (None, "KERNEL_DISPATCH"):
            continue
        wx = int(col(r, "Workgroup_Size_X", "Workgroup_Size") or 64)
        wy, wz = int(col(r, "Workgroup_Size_Y") or 1), int(col(r, "Workgroup_Size_Z") or 1)
        gx = int(col(r, "Grid_Size_X", "Grid_Size") or 64)
        gy, gz = int(col(r, "Grid_Size_Y") or 1), int(col(r, "Grid_Size_Z") or 1)
        wg = wx * wy * wz
        n_wg = max(1, (gx * gy * gz) // max(1, wg))
        waves = n_wg * math.ceil(wg / 64)
        out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), family(r["Kernel_Name"]), waves, r["Kernel_Name"]))
    out.sort()
    return out


def base_name(kernel):
    """'ancsh::pose::foo_kernel<3, true>(int, ...)' / 'void ancsh::foo_kernel<...>' -> 'foo_kernel'"""
    k = kernel.split("(")[0].strip()
    if k.startswith("void "):
        k = k[5:]
    depth, out = 0, []
    for ch in k:                                  # drop template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif depth == 0:
            out.append(ch)
    return "".join(out).split("::")[-1].strip()


def load_counters(path):
    """kernel base name -> chip-wide pipe cycles of ONE launch (all SIMDs summed): MFMA busy cycles + 4 cycles per other VALU
    instruction.  tools/sq_counters.py writes per-launch MEANS over the chip (its `launches` column = launches averaged); template
    variants of one kernel are averaged by that count."""
    work = collections.defaultdict(lambda: [0.0, 0.0])
    for r in csv.DictReader(open(path)):
        n = float(r["launches"] or 0)
        if n <= 0:
            continue
        mf, valu, mfi = float(r["SQ_VALU_MFMA_BUSY_CYCLES"] or 0), float(r["SQ_INSTS_VALU"] or 0), float(r["SQ_INSTS_MFMA"] or 0)
        w = work[base_name(r["kernel"])]
        w[0] += (mf + 4.0 * max(0.0, valu - mfi)) * n
        w[1] += n
    return {k: v[0] / v[1] for k, v in work.items() if v[1] > 0}


def account(rows, steps, trim, marker="ransac_joint_lm", work=None, clock_ghz=2.4):
    # a kernel launched exactly once per step marks the step boundaries: the window runs from the start of one marker launch to
    # the start of a later one, inside the LAST `steps` steps of the trace (the timed region; what precedes it is set-up:
    # eager warm-up runs and graph capture), with `trim` of them dropped at both ends (pipeline fill / drain)
    marks = sorted(r[0] for r in rows if marker in r[4])
    if len(marks) >= 8:
        if steps:
            marks = marks[-int(steps):]
        a, b = int(trim * len(marks)), len(marks) - 1 - int(trim * len(marks))
        lo, hi, frac_steps = marks[a], marks[b], float(b - a)
    else:
        t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
        lo, hi = t_lo + trim * (t_hi - t_lo), t_hi - trim * (t_hi - t_lo)
        frac_steps = steps * (hi - lo) / (t_hi - t_lo) if steps else None
    busy = collections.defaultdict(float)
    dur = collections.defaultdict(float)
    count = collections.defaultdict(int)
    waves_of = collections.defaultdict(list)
    pwork = collections.defaultdict(float)
    events = []
    for s, e, fam, waves, _ in rows:
        s2, e2 = max(s, lo), min(e, hi)
        if e2 <= s2:
            continue
        share = min(1.0, waves / SIMDS)
        busy[fam] += (e2 - s2) * share
        dur[fam] += e2 - s2
        count[fam] += 1
        waves_of[fam].append(waves)
        # pipe-work rate of this dispatch while it runs (SIMD-cycles per ns); unknown kernels (aten copies): a token 2 % of their share
        wk = None if work is None else work.get(base_name(_))
        rate = (wk / max(1, e - s)) if wk is not None else 0.02 * share * SIMDS * clock_ghz
        if work is not None and wk is not None:
            pwork[fam] += wk * (e2 - s2) / max(1, e - s)
        events.append((s2, 0, fam, share, rate))
        events.append((e2, 1, fam, share, rate))
    events.sort(key=lambda x: (x[0], -x[1]))
    active = collections.defaultdict(float)
    active_rate = collections.defaultdict(float)
    pipe = collections.defaultdict(float)
    attributed = collections.defaultdict(float)
    idle = over = 0.0
    hist = collections.defaultdict(float)          # time by number of concurrently running dispatches
    running = 0
    prev = lo
    for t, kind, fam, share, rate in events:
        dt = t - prev
        if dt > 0:
            R = sum(active_rate.values())
            if R > 0 and running > 0:
                for f, v in active_rate.items():
                    if v > 1e-12:
                        pipe[f] += dt * v / R
            S = sum(active.values())
            occ = min(1.0, S)
            idle += dt * (1.0 - occ)
            over += dt * max(0.0, S - 1.0)
            hist[min(running, 16)] += dt
            if S > 0:
                for f, v in active.items():
                    if v > 1e-12:
                        attributed[f] += dt * occ * v / S
            prev = t
        if kind == 0:
            active[fam] += share
            active_rate[fam] += rate
            running += 1
        else:
            active[fam] = max(0.0, active[fam] - share)
            active_rate[fam] = max(0.0, active_rate[fam] - rate)
            running -= 1
    idle += max(0.0, hi - prev)
    return dict(window_ns=hi - lo, steps=frac_steps, busy=busy, dur=dur, count=count, attributed=attributed, idle=idle, over=over,
                hist=hist, waves=waves_of, pipe=pipe if work is not None else None, pwork=pwork, clock_ghz=clock_ghz)


def report(a, out):
    n = a["steps"] or 1.0
    ms = lambda x: x / n * 1e-6
    w = out.write
    w("window %.1f ms = %.1f steps -> %.4f ms per step\n" % (a["window_ns"] * 1e-6, n, ms(a["window_ns"])))
    pipe = a.get("pipe")
    w("%-36s %9s %12s %12s %12s %12s%s\n" % ("family", "launches", "duration_ms", "busy_simd_ms", "attributed_ms", "median_waves",
                                             "      pipe_ms     floor_ms" if pipe is not None else ""))
    tot_b = tot_a = tot_p = tot_f = 0.0
    key = (lambda f: -pipe[f]) if pipe is not None else (lambda f: -a["attributed"][f])
    for fam in sorted(a["busy"], key=key):
        wv = sorted(a["waves"][fam])
        extra = ""
        if pipe is not None:
            floor = a["pwork"][fam] / (SIMDS * a["clock_ghz"])            # ns of the whole chip at 100 % issue
            extra = " %12.4f %12.4f" % (ms(pipe[fam]), ms(floor))
            tot_p += ms(pipe[fam])
            tot_f += ms(floor)
        w("%-36s %9.1f %12.4f %12.4f %12.4f %12d%s\n" % (fam, a["count"][fam] / n, ms(a["dur"][fam]), ms(a["busy"][fam]), ms(a["attributed"][fam]),
                                                          wv[len(wv) // 2], extra))
        tot_b += ms(a["busy"][fam])
        tot_a += ms(a["attributed"][fam])
    w("%-36s %9s %12s %12.4f %12.4f %12s%s\n" % ("sum", "", "", tot_b, tot_a, "", " %12.4f %12.4f" % (tot_p, tot_f) if pipe is not None else ""))
    if pipe is not None:
        w("pipe_ms: the step split by whose instructions the SIMDs were issuing (pipe work per launch from the SQ counters, kernel alone);\n"
          "floor_ms: that work at 100 %% issue on all %d SIMDs at %.2f GHz -- pipe_ms >= floor_ms for every family by construction of the split\n"
          "only if nothing else stretches it; a family whose pipe_ms is BELOW its floor_ms would imply more than the peak (the test of this table)\n"
          % (SIMDS, a["clock_ghz"]))
    w("%-36s %9s %12s %12s %12.4f   (no dispatch's waves on that share of the SIMDs)\n" % ("idle SIMD share", "", "", "", ms(a["idle"])))
    w("%-36s %9s %12s %12s %12.4f   (sum of shares above 1: dispatches time-sharing SIMDs)\n" % ("oversubscribed", "", "", "", ms(a["over"])))
    w("attributed + idle = %.4f ms per step\n" % (tot_a + ms(a["idle"])))
    w("time by number of dispatches running at once: " + ", ".join("%d: %.1f%%" % (k, 100 * v / a["window_ns"]) for k, v in sorted(a["hist"].items())) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=float, default=0, help="steps of the timed region = the last launches of the once-per-step marker kernel")
    ap.add_argument("--trim", type=float, default=0.15)
    ap.add_argument("--out", default=None)
    ap.add_argument("--counters", default=None, help="per-kernel SQ counters (tools/sq_counters.py csv): adds the pipe_ms / floor_ms columns")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    a = ap.parse_args()
    rows = load(a.trace)
    res = account(rows, a.steps, a.trim, work=load_counters(a.counters) if a.counters else None, clock_ghz=a.clock_ghz)
    out = open(a.out, "w") if a.out else sys.stdout
    out.write("# tools/step_account.py %s --steps %g --trim %g   (%d dispatches)\n" % (a.trace.split("/")[-1], a.steps, a.trim, len(rows)))
    report(res, out)


if __name__ == "__main__":
    main()
