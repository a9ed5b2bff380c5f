#!/usr/bin/env python
"""gpurun_out/<tag>/ (written by tools/capture_profiles.sh on the GPU box) -> the files committed under profiles/:
  <tag>_bench_*.json                       the bench lines (driver's exact command, rocprof'd run of it, configs[3]/[4], net only)
  <tag>_kernel_stats_bench_*.csv           rocprofv3 --kernel-trace --stats summaries of those commands
  <tag>_pmc_traffic.json (+ per-kernel csv) HBM bytes per launch per kernel family from the separate FETCH_SIZE / WRITE_SIZE passes
  <tag>_rocprof_roofline.json              per-family average launch durations FROM THE ROCPROF SUMMARY of the driver command and
                                           the roofline fractions they give (what bench.py reports as roofline.rocprof)
usage: python tools/summarise_profiles.py r02"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")

def load_bench(path):
    """stdout of a bench.py run: since round 6 the full record is printed first ({"bench_detail": {...}}) and the compact contract line last.
    -> the full record with the contract line under "contract_line" (None when the file holds neither)."""
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        return None
    last = json.loads(lines[-1])
    side = path[:-5] + "_detail.json"              # the sidecar the capture script asks bench.py for (the top-level process keeps the record off stdout)
    if os.path.exists(side):
        return dict(json.load(open(side)), contract_line=last, contract_line_bytes=len(lines[-1].rstrip("\n")))
    for l in reversed(lines):
        if l.startswith('{"bench_detail"'):
            return dict(json.loads(l)["bench_detail"], contract_line=last, contract_line_bytes=len(lines[-1].rstrip("\n")))
    return last


for name, out in (("bench_default.json", "bench_default.json"), ("bench_driver_cmd.json", "bench_driver_cmd_steps20_warmup5.json"), ("bench_default_rocprof.json", "bench_default_under_rocprof.json"),
                  ("bench_laptop_B16_N2048_K2.json", "bench_laptop_B16_N2048_K2.json"),
                  ("bench_drawer_B16_N2048_K4.json", "bench_drawer_B16_N2048_K4.json"), ("bench_net.json", "bench_net_only.json")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 10:
        rec = load_bench(p)
        if rec is None:
            shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, out)))
        else:
            with open(os.path.join(dst, "%s_%s" % (tag, out)), "w") as fh:
                json.dump(rec, fh)
                fh.write("\n")
for d, out in (("prof_default", "bench_default"), ("prof_slots1", "bench_slots1"), ("prof_laptop", "bench_laptop_B16_N2048_K2"), ("prof_drawer", "bench_drawer_B16_N2048_K4"),
               ("prof_sa_steady", "sa_steady"), ("prof_ops_beyond", "ops_beyond_L3"), ("prof_ops2048", "ops_beyond_L3_B16_N2048"),
               ("prof_ops2048_multi", "ops_multi_beyond_L3_B16_N2048")):
    p = os.path.join(src, d, "full_kernel_stats.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, out)))
for name in ("sa_steady.txt", "step_account.txt", "sq_counters_per_kernel.csv", "sq_counters_summary.txt", "ops_in_L3.json", "ops_beyond_L3.json",
             "ops_beyond_L3_B16_N2048.json", "ops_multi_beyond_L3.json", "ops_multi_beyond_L3_B16_N2048.json", "ops_fused_multi_beyond_L3_B16_N2048.json",
             "pose_tie_rate.txt", "hbm_copy_variants.txt", "ops_per_kernel_B16_N2048.txt", "mid_section.txt", "latency.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 10:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))

# ---- PMC traffic ------------------------------------------------------------------------------------------------------
traffic = os.path.join(dst, "%s_pmc_traffic.json" % tag)
if os.path.isdir(os.path.join(src, "pmc", "FETCH_SIZE")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_to_traffic.py"), os.path.join(src, "pmc"), traffic,
                           os.path.join(dst, "%s_pmc_hbm_counters_per_kernel.csv" % tag)])
    res = json.load(open(traffic))
    stamp = os.path.join(src, "source_digests.json")
    if os.path.exists(stamp):          # the sources the counters were collected on (tools/capture_profiles.sh) + their commit
        res.update(json.load(open(stamp)))
    for key, d, note in (("ops_ball_query+group_hbm_bytes_per_batch", "ops_pmc", "the five operators as five launches, one operand set (inside the Infinity Cache)"),
                         ("ops_ball_query+group_beyond_L3_hbm_bytes_per_batch", "ops_beyond_pmc", "the five operators as five launches, 12 rotating operand sets (beyond the Infinity Cache)"),
                         ("ops_fused_ball_query+group_hbm_bytes_per_batch", "ops_fused_pmc", "query_ball_group_xyz x2 + group_point(features)"),
                         ("ops_ball_query+group_beyond_L3_B16_N2048_hbm_bytes_per_batch", "ops2048_pmc", "the five operators as five launches at 16 x 2048 "
                          "(configs[3] / [4] per GPU), 12 rotating operand sets; algorithmic bytes 16*5380096 = 86081536"),
                         ("ops_multi_ball_query+group_beyond_L3_B16_N2048_hbm_bytes_per_batch", "ops2048_multi_pmc", "the same five results from two launches "
                          "(ancsh_query_ball_point_multi + ancsh_group_point_multi) at 16 x 2048, 12 rotating operand sets")):
        tot = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = glob.glob(os.path.join(src, d, c, "*counter_collection.csv"))
            if not f:
                continue
            for r in csv.DictReader(open(f[0])):
                if r["Counter_Name"] == c and ("query_ball" in r["Kernel_Name"] or "group_point" in r["Kernel_Name"] or "group_xyz" in r["Kernel_Name"]) and "fps" not in r["Kernel_Name"]:
                    tot.setdefault((r["Kernel_Name"][:60], r["Grid_Size"]), {}).setdefault(c, []).append(float(r["Counter_Value"]))
        if tot:
            res[key] = round(sum((2 * sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) + sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])) * 1024
                                 for v in tot.values() if "FETCH_SIZE" in v and "WRITE_SIZE" in v))
            res[key + "_note"] = note + (": " if "16 x 2048" in note else " at B=32, N=1024: ") + "(2*FETCH_SIZE + WRITE_SIZE)*1024 averaged per launch and summed over the launches" + ("" if "16 x 2048" in note else "; algorithmic bytes 32*5355520 = 171376640")
    json.dump(res, open(traffic, "w"), indent=1)

# ---- rocprof-derived roofline of the driver command ---------------------------------------------------------------------
stats = os.path.join(src, "prof_default", "full_kernel_stats.csv")
line = os.path.join(src, "bench_default_rocprof.json")
if os.path.exists(stats) and os.path.exists(line):
    rows = list(csv.DictReader(open(stats)))
    avg = lambda key: next((float(r["AverageNs"]) * 1e-3 for r in rows if key in r["Name"]), None)      # us
    B = 32
    sa1 = 2.0 * B * 512 * 64 * (3 * 64 + 64 * 64 + 64 * 128)
    sa2 = 2.0 * B * 128 * 64 * (3 * 128 + 128 * 128 + 128 * 256)      # executed: the first layer's feature part is summed once per source point by a conv launch
    t1, t2 = avg("sa1_fused_kernel"), avg("sa2_fused_kernel")
    # networks per launch: since round 3 the pipeline evaluates the ANCSH and the NPCS network in one grouped launch per level
    # (2 fused-SA launches per step instead of 4), i.e. twice the FLOPs per launch
    lj = load_bench(line)
    per_step = (lj.get("roofline_all", {}).get("shared_mlp_fused_sa", {}) or {}).get("launches_per_step", 4)
    G = max(1, 4 // max(1, per_step))
    sa1g, sa2g = G * sa1, G * sa2
    out = {"source": "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (%s_kernel_stats_bench_default.csv): average "
                     "duration over ALL launches of the command, i.e. mostly graph replays with 16 batches in flight (kernels of other "
                     "batches share the chip), plus the eager profiling passes" % tag,
           "sa1_fused_us": t1, "sa2_fused_us": t2,
           "networks_per_launch": G,
           "shared_mlp_fused_sa": {"achieved_TFLOPs": round((sa1g + sa2g) / ((t1 + t2) * 1e-6) / 1e12, 2),
                                   "frac": round((sa1g + sa2g) / ((t1 + t2) * 1e-6) / 1e12 / 157.3, 4)},
           "in_process_hip_events_same_run": lj.get("roofline")}
    steady = os.path.join(src, "prof_sa_steady", "full_kernel_stats.csv")
    if os.path.exists(steady):
        srows = list(csv.DictReader(open(steady)))
        savg = lambda key: next((float(r["AverageNs"]) * 1e-3 for r in srows if key in r["Name"]), None)
        s1, s2 = savg("sa1_fused_kernel"), savg("sa2_fused_kernel")
        out["sa_steady"] = {"source": "rocprofv3 --kernel-trace --stats -- python tools/sa_steady.py (%s_kernel_stats_sa_steady.csv): the same two "
                                      "launches (%d network(s) per launch, like the pipeline) alone on the chip, 2000 back-to-back each" % (tag, G),
                            "sa1_fused_us": round(s1, 1), "sa2_fused_us": round(s2, 1),
                            "achieved_TFLOPs": round((sa1g + sa2g) / ((s1 + s2) * 1e-6) / 1e12, 2),
                            "frac": round((sa1g + sa2g) / ((s1 + s2) * 1e-6) / 1e12 / 157.3, 4)}
    one = os.path.join(src, "prof_slots1", "full_kernel_stats.csv")
    if os.path.exists(one):
        orows = list(csv.DictReader(open(one)))
        oavg = lambda key: next((float(r["AverageNs"]) * 1e-3 for r in orows if key in r["Name"]), None)
        o1, o2 = oavg("sa1_fused_kernel"), oavg("sa2_fused_kernel")
        fam_us = {}
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from step_account import family
        for r in orows:
            f = family(r["Name"])
            if f:
                fam_us[f] = fam_us.get(f, 0.0) + float(r["TotalDurationNs"]) * 1e-3
        calls = next((int(r["Calls"]) for r in orows if "sa1_fused_kernel" in r["Name"]), 1)
        out["slots1"] = {"source": "rocprofv3 --kernel-trace --stats -- python bench.py --only-timed --slots 1 --steps 128 --warmup 16 (%s_kernel_stats_bench_slots1.csv): ONE "
                                   "batch in flight, so every kernel of the step runs alone on the chip in step order and its average is its exclusive duration" % tag,
                         "sa1_fused_us": round(o1, 1), "sa2_fused_us": round(o2, 1),
                         "achieved_TFLOPs": round((sa1g + sa2g) / ((o1 + o2) * 1e-6) / 1e12, 2),
                         "frac": round((sa1g + sa2g) / ((o1 + o2) * 1e-6) / 1e12 / 157.3, 4),
                         "family_us_per_step": {k: round(v / calls, 1) for k, v in sorted(fam_us.items(), key=lambda kv: -kv[1])}}
    acct = os.path.join(src, "step_account.txt")
    if os.path.exists(acct):
        last = [l for l in open(acct) if l.startswith("time by number of dispatches")]
        if last:
            out["dispatches_running_at_once_in_the_pipelined_step"] = last[-1].strip()
    json.dump(out, open(os.path.join(dst, "%s_rocprof_roofline.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1))
print(sorted(f for f in os.listdir(dst) if f.startswith(tag)))
