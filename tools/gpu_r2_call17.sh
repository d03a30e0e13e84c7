#!/bin/bash
export TMPDIR=/tmp
for env in "X=1" "FAER_HIP_LLT_FIRST=256" "FAER_HIP_LLT_FIRST=512" "X=2" "FAER_HIP_LLT_FIRST=256" "FAER_HIP_LLT_FIRST=128"; do
  timeout 200 env $env python tools/gpu_exp_one.py llt 16384 2>&1 | grep -v amdgpu
done
