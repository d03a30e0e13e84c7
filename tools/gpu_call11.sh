#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_factor.py tests/test_gpu_matmul.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
for cus in 32 16 24; do echo "panel cus $cus"; FAER_HIP_PANEL_CUS=$cus timeout 300 python tools/gpu_diag.py llt 2>&1 | tail -1; done
rm -rf gpurun_out/prof_llt11
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_llt11 -o llt -- python bench.py --workload llt --steps 3 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_llt11.log 2>&1; echo "prof rc=$?"
