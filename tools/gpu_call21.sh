#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_lu21
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lu21 -o lu -- python bench.py --workload lu --steps 2 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_lu21.log 2>&1; echo "prof rc=$?"
echo "no lookahead:"; FAER_HIP_NO_LOOKAHEAD=1 timeout 300 python tools/gpu_diag.py lu 2>&1 | tail -1
