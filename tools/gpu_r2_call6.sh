#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 env FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_leaf_phases.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2c6_leaf_phases.txt
