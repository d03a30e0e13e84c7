#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_factor.py -m gpu -q -x 2>&1 | tail -8
python tools/gpu_exp_l2.py lu 2>&1 | grep -v amdgpu
