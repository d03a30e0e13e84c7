#!/bin/bash
# LU panel kernel variants in one visit: current (in-panel look-ahead), header-only, previous
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c19
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "plu or lu_ or dist or rccl or det" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 ${O}_pytest.log
for rep in 1 2; do
for env in "X=1" "FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_hdr.so" "FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_prev.so"; do
  timeout 200 env $env python tools/gpu_exp_one.py lu 16384 2>&1 | grep -v amdgpu | cut -c1-120
done
done
for env in "X=1" "FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_hdr.so"; do
  timeout 200 env $env python tools/gpu_exp_one.py lu 4096 2>&1 | grep -v amdgpu | cut -c1-120
done
