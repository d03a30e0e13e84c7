#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c20
timeout 900 python -m pytest tests/test_gpu_hessenberg.py tests/test_gpu_bidiag.py tests/test_gpu_tridiag.py -q --tb=short -x > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -25 ${O}_pytest.log | cut -c1-300
timeout 300 python tools/gpu_bidiag_time.py 2>&1 | grep -v amdgpu
