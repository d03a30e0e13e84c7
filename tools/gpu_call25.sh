#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_factor.py -m gpu -q -x 2>&1 | tail -4
echo "lookahead:"; timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -3
echo "no lookahead:"; FAER_HIP_NO_LOOKAHEAD=1 timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -1
rm -rf gpurun_out/prof_lu25
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lu25 -o lu -- python bench.py --workload lu --steps 2 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_lu25.log 2>&1; echo "prof rc=$?"
