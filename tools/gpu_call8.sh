#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so timeout 300 python tools/gpu_diag.py lu > gpurun_out/panel_timing.log 2>&1; echo "rc=$?"
cat gpurun_out/panel_timing.log
