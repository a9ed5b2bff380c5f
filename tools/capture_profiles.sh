#!/bin/bash
# Collects the evidence committed under profiles/ for one round (run on the GPU box through gpurun):
#   tools/capture_profiles.sh r02
# 1. the driver's exact command (`python bench.py`) plain and under `rocprofv3 --kernel-trace --stats`
# 2. the single-GPU lines of configs[3] / configs[4] (16 x 2048, K = 2 / 4) plain + rocprof stats
# 2b. tools/sa_steady.py (the fused SA launches alone, back-to-back) plain + rocprof stats
# 3. HBM traffic: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE passes (never combined with other trace domains)
#    for the end-to-end step and for the five-operator ball-query + group graph
# Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/ (tools/summarise_profiles.py does).
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o full -- python $ROOT/bench.py --no-cpu-baseline > $O/bench_default_rocprof.json 2> $O/bench_default_rocprof.err
for cfg in "laptop 2" "drawer 4"; do
  set -- $cfg
  python $ROOT/bench.py --batch 16 --npoints 2048 --parts $2 --no-cpu-baseline > $O/bench_$1_B16_N2048_K$2.json 2> $O/bench_$1.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o full -- python $ROOT/bench.py --batch 16 --npoints 2048 --parts $2 --no-cpu-baseline --steps 128 > $O/bench_$1_rocprof.json 2> $O/bench_$1_rocprof.err
done
python $ROOT/bench.py --workload net --no-cpu-baseline > $O/bench_net.json 2> $O/bench_net.err
# the roofline's kernels alone on the chip at the loaded clock (2000 back-to-back launches each)
python $ROOT/tools/sa_steady.py > $O/sa_steady.txt 2> $O/sa_steady.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sa_steady -o full -- python $ROOT/tools/sa_steady.py > $O/sa_steady_rocprof.txt 2> $O/sa_steady_rocprof.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/$C -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --slots 1 --no-graph > $O/pmc_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ops_pmc/$C -o pmc -- python $ROOT/tools/ops_bench.py > $O/ops_pmc_$C.log 2>&1
  FUSED=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ops_fused_pmc/$C -o pmc -- python $ROOT/tools/ops_bench.py > $O/ops_fused_pmc_$C.log 2>&1
done
# keep the merge-back small: stats + counter tables only (the raw kernel traces are tens of MB)
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*.db" -delete
ls -R $O | head -60
cut -c1-400 $O/bench_default.json
