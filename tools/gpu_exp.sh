#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/exp.log
: > $O
python tools/gpu_micro_syrk.py 2>&1 | grep -v amdgpu.ids | grep -E "dgemm|r=15360|r=8192" >> $O
run() { timeout 240 env "$@" python tools/gpu_exp_l2.py $WHAT $N >> $O 2>&1 || echo "FAILED: $* $WHAT" >> $O; }
N=16384
WHAT=lu
run X=1
WHAT=llt
run X=1
run FAER_HIP_LLT_TAIL=0
run FAER_HIP_LLT_TAIL=6144
grep -v amdgpu.ids $O
rm -rf gpurun_out/trace_llt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_llt -o llt -- python tools/gpu_exp_l2.py llt > gpurun_out/trace_llt.log 2>&1; echo "trace llt rc=$?"
timeout 800 python -m pytest tests/test_gpu_matmul.py tests/test_gpu_factor.py -m gpu -q -x 2>&1 | tail -3
