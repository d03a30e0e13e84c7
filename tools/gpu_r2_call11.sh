#!/bin/bash
# fused solve+update in the LU panel recursion: parity subset, then A/B inside one visit
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c11
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "plu or lu_solve or lu_ or dist or rccl or fplu or det" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 ${O}_pytest.log
for env in "X=1" "FAER_HIP_LU_FUSE=0" "X=2" "FAER_HIP_LU_FUSE=0"; do
  echo "== $env"
  timeout 200 env $env python tools/gpu_exp_one.py lu 16384 2>&1 | grep -v amdgpu
done
for env in "X=1" "FAER_HIP_LU_FUSE=0"; do
  echo "== $env (8192)"
  timeout 200 env $env python tools/gpu_exp_one.py lu 8192 2>&1 | grep -v amdgpu
done
