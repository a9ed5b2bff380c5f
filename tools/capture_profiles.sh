#!/bin/bash
# Collects the evidence committed under profiles/ for one round (run on the GPU box through gpurun):
#   tools/capture_profiles.sh <tag> <commit> [sections...]        e.g.  tools/capture_profiles.sh r03 abc1234 bench account sq
# sections (default: all)
#   bench    `python bench.py` (512 steps: steady state)
#   driver   the driver's exact command of rounds 1-2: `python3 bench.py --gpus 1 --steps 20 --warmup 5` (pipeline fill + drain included)
#   rocprof  the same under `rocprofv3 --kernel-trace --stats`
#   rocprof1 `bench.py --only-timed --slots 1` under `rocprofv3 --kernel-trace --stats`: every kernel alone on the chip, in step order
#   account  `bench.py --only-timed` under `rocprofv3 --kernel-trace` -> tools/step_account.py (occupancy-weighted account of a step)
#   configs  the single-GPU lines of configs[3] / configs[4] (16 x 2048, K = 2 / 4) plain + rocprof stats
#   net      configs[1]: network only
#   steady   tools/sa_steady.py (the fused SA launches alone, back-to-back) plain + rocprof stats
#   pmc      HBM traffic of the step's kernels: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE passes
#   ops      the five-operator ball-query + group graph: in the Infinity Cache (1 operand set) and beyond it (12 sets), time + PMC
#   ops2048  the same graph at the configs[3] / [4] shape (16 x 2048): five launches and the two-launch form, time + per-kernel trace + PMC
#   tie      tools/pose_tie_rate.py at the reference's 10000 / 200 budgets: HIP pose fit vs the reference arithmetic on replayed draws (how often the consensus sets differ, by how much)
#   mid      tools/mid_bench.py: the backbone's mid-section, layer-by-layer launches against the chain launches, isolated
#   latency  bench.py --latency-leg: one cloud, one slot
#   copy     the float4-copy HBM yardstick of bench.py in its three variants
#   sq       SQ counters per kernel (MFMA instructions / busy cycles, CU busy cycles, wave cycles) in separate passes
# PMC passes are never combined with any trace domain other than --kernel-trace.
# Everything lands in gpurun_out/<tag>/; tools/summarise_profiles.py <tag> copies what is to be judged into profiles/.
TAG=${1:-r05}
COMMIT=${2:-unknown}
shift 2
SECTIONS=${*:-bench driver rocprof rocprof1 account configs net steady pmc ops ops2048 tie copy sq mid latency}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
if has pmc; then   # the stamp belongs to the counters: written only by the run that collects them
python - > $O/source_digests.json <<PY
import json, sys
sys.path.insert(0, "$ROOT")
import bench
print(json.dumps({"commit": "$COMMIT", "source_digests": bench.source_digests()}))
PY
fi
if has bench; then
  ANCSH_BENCH_DETAIL=$O/bench_default_detail.json python $ROOT/bench.py > $O/bench_default.json 2> $O/bench_default.err
  cut -c1-600 $O/bench_default.json
fi
if has driver; then
  ANCSH_BENCH_DETAIL=$O/bench_driver_cmd_detail.json python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
  cut -c1-300 $O/bench_driver_cmd.json
fi
if has rocprof; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o full -- env ANCSH_BENCH_DETAIL=$O/bench_default_rocprof_detail.json python $ROOT/bench.py --no-cpu-baseline > $O/bench_default_rocprof.json 2> $O/bench_default_rocprof.err
fi
if has rocprof1; then
  # the step's kernels ALONE on the chip in step order (one batch in flight: no other batch's kernels time-share the SIMDs), so that the
  # rocprofv3 average of a kernel is its exclusive duration -- with 20 batches in flight 8-10 dispatches overlap and every average is stretched
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slots1 -o full -- python $ROOT/bench.py --only-timed --slots 1 --steps 128 --warmup 16 > $O/bench_slots1_rocprof.json 2> $O/bench_slots1_rocprof.err
  cut -c1-200 $O/bench_slots1_rocprof.json
fi
if has account; then
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_account -o acct -- python $ROOT/bench.py --only-timed --steps 256 --warmup 32 > $O/bench_only_timed_rocprof.json 2> $O/bench_only_timed_rocprof.err
  T=$(find $O/prof_account -name "*kernel_trace.csv" | head -1)
  # pipe-work-weighted attribution needs this round's SQ counters (section sq): run `account` after `sq`, or the previous round's table is used
  CNT=$O/sq_counters_per_kernel.csv; [ -f $CNT ] || CNT=$(ls $ROOT/profiles/*_sq_counters_per_kernel.csv | tail -1)
  python $ROOT/tools/step_account.py $T --steps 256 --trim 0.2 --counters $CNT --out $O/step_account.txt
  gzip -c $T > $O/account_kernel_trace.csv.gz
  cat $O/step_account.txt $O/bench_only_timed_rocprof.json
fi
if has configs; then
  for cfg in "laptop 2" "drawer 4"; do
    set -- $cfg
    ANCSH_BENCH_DETAIL=$O/bench_$1_B16_N2048_K$2_detail.json python $ROOT/bench.py --batch 16 --npoints 2048 --parts $2 --no-cpu-baseline > $O/bench_$1_B16_N2048_K$2.json 2> $O/bench_$1.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o full -- python $ROOT/bench.py --batch 16 --npoints 2048 --parts $2 --no-cpu-baseline --steps 128 > $O/bench_$1_rocprof.json 2> $O/bench_$1_rocprof.err
  done
fi
if has net; then
  ANCSH_BENCH_DETAIL=$O/bench_net_detail.json python $ROOT/bench.py --workload net --no-cpu-baseline > $O/bench_net.json 2> $O/bench_net.err
fi
if has steady; then
  # the roofline's kernels alone on the chip at the loaded clock (2000 back-to-back launches each)
  python $ROOT/tools/sa_steady.py > $O/sa_steady.txt 2> $O/sa_steady.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sa_steady -o full -- python $ROOT/tools/sa_steady.py > $O/sa_steady_rocprof.txt 2> $O/sa_steady_rocprof.err
fi
STEP="python $ROOT/bench.py --steps 2 --warmup 1 --slots 1 --no-graph --only-timed"
if has pmc; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/$C -o pmc -- $STEP > $O/pmc_$C.log 2>&1
  done
fi
if has ops; then
  python $ROOT/tools/ops_bench.py --sets 1 > $O/ops_in_L3.json 2> $O/ops_in_L3.err
  python $ROOT/tools/ops_bench.py --sets 12 > $O/ops_beyond_L3.json 2> $O/ops_beyond_L3.err
  python $ROOT/tools/ops_bench.py --sets 12 --batch 16 --npoints 2048 > $O/ops_beyond_L3_B16_N2048.json 2>> $O/ops_beyond_L3.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ops_beyond -o full -- python $ROOT/tools/ops_bench.py --sets 12 > $O/ops_beyond_L3_rocprof.json 2> $O/ops_beyond_L3_rocprof.err
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ops_pmc/$C -o pmc -- python $ROOT/tools/ops_bench.py --sets 1 --reps 4 > $O/ops_pmc_$C.log 2>&1
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ops_beyond_pmc/$C -o pmc -- python $ROOT/tools/ops_bench.py --sets 12 --reps 2 > $O/ops_beyond_pmc_$C.log 2>&1
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ops_fused_pmc/$C -o pmc -- python $ROOT/tools/ops_bench.py --sets 1 --reps 4 --mode fused > $O/ops_fused_pmc_$C.log 2>&1
  done
  cat $O/ops_in_L3.json $O/ops_beyond_L3.json | cut -c1-400
fi
if has ops2048; then
  A="--sets 12 --batch 16 --npoints 2048"
  python $ROOT/tools/ops_bench.py $A > $O/ops_beyond_L3_B16_N2048.json 2> $O/ops2048.err
  python $ROOT/tools/ops_bench.py $A --mode multi > $O/ops_multi_beyond_L3_B16_N2048.json 2>> $O/ops2048.err
  python $ROOT/tools/ops_bench.py $A --mode fused_multi > $O/ops_fused_multi_beyond_L3_B16_N2048.json 2>> $O/ops2048.err
  python $ROOT/tools/ops_bench.py --sets 12 --mode multi > $O/ops_multi_beyond_L3.json 2>> $O/ops2048.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ops2048 -o full -- python $ROOT/tools/ops_bench.py $A > /dev/null 2>> $O/ops2048.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ops2048_multi -o full -- python $ROOT/tools/ops_bench.py $A --mode multi > /dev/null 2>> $O/ops2048.err
  python - > $O/ops_per_kernel_B16_N2048.txt <<PY
import csv, glob, collections
print("per-launch durations (us, median over the graph replays) of the op-level ball_query + group graph at 16 x 2048, 12 rotating operand sets,")
print("from rocprofv3 --kernel-trace (inside a graph a kernel's start stamp follows the previous kernel's end: a duration includes its launch gap)")
for d, what in (("prof_ops2048", "five launches (the reference's operator sequence)"), ("prof_ops2048_multi", "two launches (ancsh_query_ball_point_multi + ancsh_group_point_multi)")):
    f = glob.glob("$O/%s/*kernel_trace.csv" % d)
    if not f:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "query_ball" in k or "group_" in k:
            per[(k.split("(")[0][-44:], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(what)
    tot = 0.0
    for k, v in per.items():
        v = sorted(v); med = v[len(v) // 2]; tot += med
        print("  %-46s grid %8s x %-3s launches %5d  median %6.2f us" % (k[0], k[1], k[2], len(v), med))
    print("  sum of medians %.2f us" % tot)
PY
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ops2048_pmc/$C -o pmc -- python $ROOT/tools/ops_bench.py $A --reps 2 > $O/ops2048_pmc_$C.log 2>&1
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ops2048_multi_pmc/$C -o pmc -- python $ROOT/tools/ops_bench.py $A --reps 2 --mode multi > $O/ops2048_multi_pmc_$C.log 2>&1
  done
  cat $O/ops_per_kernel_B16_N2048.txt; cut -c1-200 $O/ops_beyond_L3_B16_N2048.json $O/ops_multi_beyond_L3_B16_N2048.json
fi
if has tie; then
  # round 5: at the REFERENCE'S budgets (evaluation/parallel_ancsh_pose.py:262,299); round 4 ran 700 / 200 / 200 clouds at 2000 / 64
  python $ROOT/tools/pose_tie_rate.py --clouds 208 --parts 3 --npoints 1024 --na 10000 --nb 200 --out $O/tie_K3.txt > $O/tie_K3.log 2>&1
  python $ROOT/tools/pose_tie_rate.py --clouds 64 --parts 4 --npoints 2048 --first 7000 --na 10000 --nb 200 --out $O/tie_K4.txt > $O/tie_K4.log 2>&1
  python $ROOT/tools/pose_tie_rate.py --clouds 64 --parts 2 --npoints 2048 --first 8000 --na 10000 --nb 200 --out $O/tie_K2.txt > $O/tie_K2.log 2>&1
  cat $O/tie_K3.txt $O/tie_K4.txt $O/tie_K2.txt > $O/pose_tie_rate.txt
  grep -E "fits|consensus" $O/pose_tie_rate.txt
fi
if has mid; then
  python $ROOT/tools/mid_bench.py > $O/mid_section.txt 2> $O/mid_section.err
  cat $O/mid_section.txt
fi
if has latency; then
  python $ROOT/bench.py --latency-leg > $O/latency.json 2> $O/latency.err
  cut -c1-400 $O/latency.json
fi
if has copy; then
  for v in 0 1 2; do ANCSH_COPY_VARIANT=$v python $ROOT/tools/hbm_copy_variants.py 2>/dev/null; done > $O/hbm_copy_variants.txt
  cat $O/hbm_copy_variants.txt
fi
if has sq; then
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq/mfma -o pmc -- $STEP > $O/sq_mfma.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq/valu -o pmc -- $STEP > $O/sq_valu.log 2>&1
  python $ROOT/tools/sq_counters.py $O/sq $O/sq_counters_per_kernel.csv $O/sq_counters_summary.txt
fi
# keep the merge-back small: stats + counter tables only (the raw kernel traces are tens of MB)
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*.db" -delete
ls $O
