#!/bin/bash
export TMPDIR=/tmp
for env in "X=1" "FAER_HIP_LLT_TAIL=2048" "FAER_HIP_LLT_TAIL=4096" "FAER_HIP_LLT_FIRST=256" "X=2"; do
  timeout 200 env $env python tools/gpu_exp_one.py llt 16384 2>&1 | grep -v amdgpu | cut -c1-110
done
for env in "X=1" "FAER_HIP_PANEL_CUS=24" "FAER_HIP_PANEL_CUS=40" "X=2"; do
  timeout 200 env $env python tools/gpu_exp_one.py lu 16384 2>&1 | grep -v amdgpu | cut -c1-110
done
