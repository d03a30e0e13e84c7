#!/bin/bash
# LLT driver knobs with the 4-wave leaf, A/B inside one visit
export TMPDIR=/tmp
mkdir -p gpurun_out
for env in "X=1" "FAER_HIP_LLT_TAIL=1024" "FAER_HIP_LLT_TAIL=2048" "FAER_HIP_LLT_TAIL=3072" "FAER_HIP_LLT_DPANEL=4096" "FAER_HIP_LLT_DPANEL=100000" "FAER_HIP_LLT_NB2=2048" "X=2"; do
  timeout 200 env $env python tools/gpu_exp_one.py llt 16384 2>&1 | grep -v amdgpu
done
for env in "X=1" "FAER_HIP_LLT_LA_MIN=100000"; do
  timeout 200 env $env python tools/gpu_exp_one.py llt 8192 2>&1 | grep -v amdgpu
done
