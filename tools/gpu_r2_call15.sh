#!/bin/bash
# kernel traces of LU and LLT (N = 16384) with the 4-wave leaf: per-queue busy / idle analysis
export TMPDIR=/tmp
mkdir -p gpurun_out
for wl in lu llt; do
  rm -rf gpurun_out/trace_$wl
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$wl -o $wl -- python bench.py --workload $wl --steps 3 --warmup 2 --no-extras --no-cpu > gpurun_out/trace_$wl.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/trace_$wl.log
  f=$(find gpurun_out/trace_$wl -name "*kernel_trace.csv" | head -1)
  python tools/trace_timeline.py $f > gpurun_out/r2c15_timeline_$wl.txt 2>&1
  rm -rf gpurun_out/trace_$wl
done
