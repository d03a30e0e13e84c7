#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_matmul.py tests/test_gpu_qr.py -m gpu -q -x 2>&1 | tail -3
for e in 1 0; do
  if [ $e = 0 ]; then export FAER_HIP_NO_SKINNY=1; fi
  python bench.py --workload qr --no-cpu --no-extras 2>&1 | grep -o "ms_per_step[^,]*"
done
unset FAER_HIP_NO_SKINNY
rm -rf gpurun_out/trace_qr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_qr -o qr -- python bench.py --workload qr --no-cpu --no-extras --steps 3 --warmup 1 > gpurun_out/trace_qr.log 2>&1; echo "trace qr rc=$?"
