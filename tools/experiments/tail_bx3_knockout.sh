#!/bin/bash
# Timing-only knock-out builds of the bf16x3 tail kernel (results wrong by construction): which part of its time is the exact bf16 split
# of the epilogues, the weight stream from L2, the input gather.  Builds scratch/ko/libancsh_<variant>.so; run each with
#   ANCSH_HIP_LIB=scratch/ko/libancsh_<variant>.so rocprofv3 --kernel-trace --stats -- python bench.py --only-timed --bf16x3 --slots 1 --steps 64
set -e
cd "$(dirname "$0")/../.."
C=articulated-pose_amd/csrc
mkdir -p scratch/ko
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000"
for v in base SPLIT WLOAD INPUT "SPLIT -DBX3_KO_WLOAD -DBX3_KO_INPUT"; do
  name=$(echo $v | tr -d ' ' | sed 's/-DBX3_KO_/_/g')
  def=""; [ "$v" != base ] && def="-DBX3_KO_$v"
  /opt/rocm/bin/hipcc $FLAGS $def -c $C/tail_bf16x3.hip -o scratch/ko/tail_$name.o &
done
wait
for o in scratch/ko/tail_*.o; do
  name=$(basename $o .o | sed 's/tail_//')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v tail_bf16x3) $o -o scratch/ko/libancsh_$name.so
done
ls scratch/ko/*.so
