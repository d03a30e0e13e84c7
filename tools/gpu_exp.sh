#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/exp.log
: > $O
run() { timeout 240 env "$@" python tools/gpu_exp_l2.py $WHAT $N >> $O 2>&1 || echo "FAILED: $* $WHAT" >> $O; }
WHAT=llt
N=16384
run FAER_HIP_LLT_NB2=1024
run FAER_HIP_LLT_NB2=2048
run FAER_HIP_LLT_NB2=1536
run FAER_HIP_LLT_NB2=2048 FAER_HIP_LLT_TAIL=6144
run FAER_HIP_LLT_NB2=1024
run FAER_HIP_LLT_NB2=2048 FAER_HIP_LLT_DPANEL=0
N=12288
run FAER_HIP_LLT_NB2=1024
run FAER_HIP_LLT_NB2=2048
grep -v amdgpu.ids $O
