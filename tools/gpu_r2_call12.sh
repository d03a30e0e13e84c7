#!/bin/bash
# 4-wave substitution leaf (16 / 32 right-hand sides per wavefront): parity subset, phase table, A/B inside one visit
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c12
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "trsm or triangular or stability or plu or llt or lu_solve or dist or rccl or qr_solve or ldlt or inverse" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 ${O}_pytest.log
timeout 300 env FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_leaf_phases.py 2>&1 | grep -v amdgpu | grep -A1 "float64 n=128 k=\(64\|8192\|16384\) left" | tee ${O}_leaf_phases.txt
for env in "X=1" "FAER_HIP_TRSM_RW=32" "FAER_HIP_TRSM_RW=16" "X=2"; do
  echo "== $env"
  timeout 200 env $env python tools/gpu_exp_one.py llt 16384 2>&1 | grep -v amdgpu
  timeout 200 env $env python tools/gpu_exp_one.py lu 16384 2>&1 | grep -v amdgpu
done
