#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
echo "lookahead:"; timeout 300 python tools/gpu_diag.py lu qr 2>&1 | tail -4
echo "no lookahead:"; FAER_HIP_NO_LOOKAHEAD=1 timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -1
