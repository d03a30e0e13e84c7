#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/exp.log
: > $O
run() { timeout 240 env "$@" python tools/gpu_exp_l2.py $WHAT $N >> $O 2>&1 || echo "FAILED: $* $WHAT" >> $O; }
N=16384
WHAT=lu
run FAER_HIP_LU_SPLIT=0
run FAER_HIP_LU_SPLIT=1
run FAER_HIP_LU_SPLIT=0
run FAER_HIP_LU_SPLIT=1
N=8192
run FAER_HIP_LU_SPLIT=0
run FAER_HIP_LU_SPLIT=1
grep -v amdgpu.ids $O
rm -rf gpurun_out/trace_lu
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_lu -o lu -- python tools/gpu_exp_l2.py lu > gpurun_out/trace_lu.log 2>&1; echo "trace lu rc=$?"
timeout 800 python -m pytest tests/test_gpu_factor.py tests/test_gpu_qr.py -m gpu -q -x 2>&1 | tail -3
