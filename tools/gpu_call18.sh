#!/bin/bash
export TMPDIR=/tmp
run() { local fails=0; for i in $(seq 1 $2); do timeout 120 python -m pytest tests/test_gpu_factor.py -m gpu -q -x -k "llt_host_pointer or llt_full_size or test_llt_solve" > /tmp/o.log 2>&1 || { fails=$((fails+1)); grep -o "NonPositivePivot { index: [0-9]* }" /tmp/o.log | head -1; }; done; echo "$1: $fails failures of $2"; }
run default 14
FAER_HIP_LA_HOSTSYNC=1 run hostsync 10
FAER_HIP_NO_CUMASK=1 run nocumask 10
