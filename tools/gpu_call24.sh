#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_qr24
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_qr24 -o qr -- python bench.py --workload qr --steps 3 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_qr24.log 2>&1; echo "prof rc=$?"
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_qr24.log
