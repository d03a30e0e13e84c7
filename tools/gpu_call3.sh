#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/pytest_gpu.log
timeout 300 python tools/gpu_diag.py llt lu qr > gpurun_out/diag_call4.log 2>&1; echo "diag rc=$?"
cat gpurun_out/diag_call4.log
rm -rf gpurun_out/prof_llt4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_llt4 -o llt -- python bench.py --workload llt --steps 3 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_llt4.log 2>&1; echo "prof rc=$?"
grep '"metric"' gpurun_out/prof_llt4.log | cut -c1-400
