#!/bin/bash
# A/B of library builds in one visit: gpurun -- 'bash tools/gpu_r4_ab_libs.sh <tag> <what> <n> lib1.so lib2.so ...'  (libs under tools/ab/)
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1; what=$2; n=$3; shift 3
O=gpurun_out/${tag}.log
: > $O
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib" >> $O
    timeout 240 env FAER_HIP_LIB=$PWD/tools/ab/$lib python tools/gpu_exp_one.py $what $n >> $O 2>&1 || echo "FAILED: $lib" >> $O
  done
done
grep "==\| n=" $O | paste - - | sed 's/residual.*//'
