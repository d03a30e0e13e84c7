#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c7
timeout 300 env FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_leaf_phases.py 2>&1 | grep -v amdgpu | grep -A1 "float64" | tee ${O}_leaf_phases.txt
timeout 600 python -m pytest tests -m gpu -q --tb=short -x -k "rccl or two_ranks or (test_trsm and not stable) or plu_vs" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -8 ${O}_pytest.log
