#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_qr.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python tools/gpu_diag.py qr 2>&1 | tail -1
for cus in 32 24 16; do echo "llt panel cus $cus"; FAER_HIP_PANEL_CUS=$cus timeout 300 python tools/gpu_diag.py llt 2>&1 | tail -1; done
rm -rf gpurun_out/prof_qr27
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_qr27 -o qr -- python bench.py --workload qr --steps 3 --warmup 1 --no-extras --no-cpu > gpurun_out/prof_qr27.log 2>&1; echo "prof rc=$?"
