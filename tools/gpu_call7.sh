#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
timeout 300 python tools/gpu_diag.py llt lu qr > gpurun_out/diag_call9.log 2>&1; echo "diag rc=$?"
cat gpurun_out/diag_call9.log
rm -rf gpurun_out/prof_lu9
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lu9 -o lu -- python bench.py --workload lu --steps 2 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_lu9.log 2>&1; echo "prof rc=$?"
