#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_qr.py -m gpu -q -x -k "colpiv" 2>&1 | tail -25
