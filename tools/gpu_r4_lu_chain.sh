#!/bin/bash
# A/B of the bulk-stream chain grouping of the LU look-ahead driver (FAER_HIP_LU_CHAIN / _ROWS), one visit.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${1:-r4chain}.log
: > $O
run() { timeout 240 env "$@" python tools/gpu_exp_one.py lu $N >> $O 2>&1 || echo "FAILED: $* n=$N" >> $O; }
for N in 8192 16384; do
  run FAER_HIP_LU_CHAIN=0
  run FAER_HIP_LU_CHAIN=1
  run FAER_HIP_LU_CHAIN=2
  run FAER_HIP_LU_CHAIN=3 FAER_HIP_LU_CHAIN_ROWS=6144
  run FAER_HIP_LU_CHAIN=3 FAER_HIP_LU_CHAIN_ROWS=8192
  run FAER_HIP_LU_CHAIN=3 FAER_HIP_LU_CHAIN_ROWS=10240
  run FAER_HIP_LU_CHAIN=3 FAER_HIP_LU_CHAIN_ROWS=12288
  run FAER_HIP_LU_CHAIN=0
done
grep "lu n=" $O
timeout 600 python -m pytest tests/test_gpu_factor.py -q -x -k "plu or lookahead_paths_fp32" 2>&1 | tail -3
