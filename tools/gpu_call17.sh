#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_factor.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -2
echo "no lookahead:"; FAER_HIP_NO_LOOKAHEAD=1 timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -1
rm -rf gpurun_out/prof_lu17
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lu17 -o lu -- python bench.py --workload lu --steps 2 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_lu17.log 2>&1; echo "prof rc=$?"
