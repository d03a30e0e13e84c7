#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/pytest_gpu.log
timeout 300 python tools/gpu_diag.py gemm llt lu > gpurun_out/diag_call2.log 2>&1; echo "diag rc=$?"
cat gpurun_out/diag_call2.log
