#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_factor.py -m gpu -q -x -k llt > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
for cus in 32 16 8; do echo "panel cus $cus"; FAER_HIP_PANEL_CUS=$cus timeout 300 python tools/gpu_diag.py llt 2>&1 | tail -3; done
FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so FAER_HIP_NO_LOOKAHEAD=1 timeout 300 python tools/gpu_diag.py llt 2>&1 | grep -v amdgpu | tail -4 > gpurun_out/leaf_timing.log; cat gpurun_out/leaf_timing.log
