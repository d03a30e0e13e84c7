#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_qr.py -m gpu -q -x 2>&1 | tail -3
for lib in tools/ab/libfaer_hip_7729917.so faer-rs_amd/libfaer_hip.so; do
  FAER_HIP_LIB=$PWD/$lib python bench.py --workload qr --no-cpu --no-extras 2>&1 | grep -o "ms_per_step[^,]*"
done
timeout 200 python tools/gpu_stress_lu.py 4096 2>&1 | grep -v amdgpu.ids | tail -4
rm -rf gpurun_out/trace_qr
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_qr -o qr -- python bench.py --workload qr --no-cpu --no-extras --steps 3 --warmup 1 > gpurun_out/trace_qr.log 2>&1; echo "trace qr rc=$?"
head -8 gpurun_out/trace_qr/qr_kernel_stats.csv | cut -c1-160
