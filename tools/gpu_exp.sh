#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_factor.py -m gpu -q -x -k "plu or lu_" 2>&1 | tail -2
for lib in tools/ab/libfaer_hip_head.so faer-rs_amd/libfaer_hip.so; do
  echo "== $lib"
  FAER_HIP_LIB=$PWD/$lib python tools/gpu_exp_l2.py lu 2>&1 | grep -v amdgpu
  FAER_HIP_LIB=$PWD/$lib python tools/gpu_size_sweep.py 256,1024,4096 2>&1 | grep -v amdgpu | tail -3
done
