#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_factor.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -3
