#!/bin/bash
# shared interchange lists (LU) and pack-less multi-block TRSM: parity subset, A/B inside one visit
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c16
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "trsm or triangular or stability or plu or lu_ or dist or rccl or qr_solve or inverse or det" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -4 ${O}_pytest.log
for env in "X=1" "FAER_HIP_LU_LISTS=0" "FAER_HIP_TRSM_PACK=1" "X=2"; do
  timeout 200 env $env python tools/gpu_exp_one.py lu 16384 2>&1 | grep -v amdgpu
done
for env in "X=1" "FAER_HIP_TRSM_PACK=1"; do
  timeout 200 env $env python tools/gpu_exp_one.py lu 8192 2>&1 | grep -v amdgpu
done
