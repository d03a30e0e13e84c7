#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/fio.log
: > $O
for c in 0 1 0 1; do
  timeout 200 env FAER_HIP_GEMM_FASTIO=$c python tools/gpu_micro_upd.py >> $O 2>&1 || echo "FAILED $c" >> $O
done
for what in llt lu; do for c in 0 1 0 1; do
  timeout 200 env FAER_HIP_GEMM_FASTIO=$c python tools/gpu_exp_one.py $what 16384 >> $O 2>&1 || echo "FAILED $what $c" >> $O
done; done
grep "FASTIO" $O | grep -v "r=4096" | sed 's/checksum.*//'
timeout 1500 python -m pytest tests/test_gpu_matmul.py tests/test_gpu_factor.py -m gpu -x -q 2>&1 | tail -4
