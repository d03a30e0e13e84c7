#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extras.py -m gpu -q -x 2>&1 | tail -25
