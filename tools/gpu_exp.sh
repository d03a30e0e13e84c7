#!/bin/bash
export TMPDIR=/tmp
python tools/gpu_size_sweep.py 1024,2048,4096,8192 2>&1 | grep -v amdgpu.ids | tail -4
python bench.py --workload qr --no-cpu --no-extras 2>&1 | grep -o "ms_per_step[^,]*"
python bench.py --workload llt --no-cpu --no-extras 2>&1 | grep -o "ms_per_step[^,]*"
python bench.py --workload lu --no-cpu --no-extras 2>&1 | grep -o "ms_per_step[^,]*"
timeout 600 python -m pytest tests/test_gpu_qr.py tests/test_gpu_matmul.py -m gpu -q -x 2>&1 | tail -2
