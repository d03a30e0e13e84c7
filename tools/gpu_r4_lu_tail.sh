#!/bin/bash
# plain recursion on the whole chip against the look-ahead driver at the sizes of an LU tail, one visit
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${1:-r4tail}.log
: > $O
for n in 3072 4096 5120 6144 8192 10240; do
  for rep in 1 2; do
    timeout 240 python tools/gpu_exp_one.py lu $n >> $O 2>&1
    timeout 240 env FAER_HIP_NO_LOOKAHEAD=1 python tools/gpu_exp_one.py lu $n >> $O 2>&1
  done
done
grep "lu n=" $O | sed 's/residual.*//'
