#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_lu6
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lu6 -o lu -- python bench.py --workload lu --steps 2 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_lu6.log 2>&1; echo "prof rc=$?"
grep '"metric"' gpurun_out/prof_lu6.log | cut -c1-300
