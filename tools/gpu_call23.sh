#!/bin/bash
export TMPDIR=/tmp
echo "diag lookahead:"; timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -1
echo "diag no lookahead:"; FAER_HIP_NO_LOOKAHEAD=1 timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -1
echo "bench lookahead:"; timeout 300 python bench.py --workload lu --steps 3 --warmup 1 --no-extras --no-cpu 2>&1 | grep -o '"ms_per_step": [0-9.]*'
echo "bench under rocprof:"; rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o lu -- python bench.py --workload lu --steps 3 --warmup 1 --no-extras --no-cpu 2>&1 | grep -o '"ms_per_step": [0-9.]*'
echo "diag lookahead nocumask:"; FAER_HIP_NO_CUMASK=1 timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -1
