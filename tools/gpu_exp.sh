#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extras.py -m gpu -q -x -k "full_piv or colpiv" 2>&1 | tail -12
