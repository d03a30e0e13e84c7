#!/bin/bash
export TMPDIR=/tmp
run() { local fails=0; for i in $(seq 1 $2); do timeout 120 python -m pytest tests/test_gpu_factor.py -m gpu -q -x -k "llt_host_pointer or llt_full_size or test_llt_solve" > /tmp/o.log 2>&1 || { fails=$((fails+1)); grep -o "NonPositivePivot { index: [0-9]* }\|Error.*" /tmp/o.log | head -1; }; done; echo "$1: $fails failures of $2"; }
run default 20
timeout 900 python -m pytest tests/test_gpu_factor.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/gpu_diag.py llt lu 2>&1 | tail -6
