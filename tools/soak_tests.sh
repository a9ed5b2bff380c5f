#!/bin/bash
# Flake hunt (run on the GPU box through gpurun, or here for the CPU suite): the same pytest selection N times, each in a FRESH process --
# the two flakes of round 6 (a late clone racing with the next graph replay; a rendezvous port inside the ephemeral range) only showed in
# fresh processes, 1 run in 30.  Usage:
#   tools/soak_tests.sh N [pytest args...]        e.g.  tools/soak_tests.sh 40 tests/test_fullsize_gpu.py -m gpu -k invariant
#                                                       tools/soak_tests.sh 6 tests -m gpu        tools/soak_tests.sh 8 tests -m "not gpu"
# Prints one line per run and, at the end, every FAILED test id with its count; output of the failing runs stays in gpurun_out/soak/.
N=${1:-10}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/soak; mkdir -p $O; rm -f $O/run*.txt
cd $ROOT
bad=0
for i in $(seq 1 $N); do
  if timeout 1800 python -m pytest "$@" -q --tb=short -p no:cacheprovider > $O/run$i.txt 2>&1; then tail -1 $O/run$i.txt; rm -f $O/run$i.txt
  else bad=$((bad + 1)); echo "run $i FAILED (rc $?): $(tail -1 $O/run$i.txt)"; fi
done
echo "$bad of $N runs failed"
grep -h "^FAILED" $O/run*.txt 2>/dev/null | sort | uniq -c
exit $((bad > 0))
