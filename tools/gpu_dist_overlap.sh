#!/bin/bash
# panel / update overlap inside a rank of the distributed LU: timeline of one rank, and the
# two-stream vs one-stream schedule with one and with two processes sharing the GPU
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r02_dist_overlap.txt
: > $O
for env in "X=1" "FAER_HIP_DIST_ONE_STREAM=1"; do
  timeout 300 env $env python tools/gpu_dist_overlap.py 8192 512 3 2>&1 | grep "dist lu" | tee -a $O
done
for env in "X=1" "FAER_HIP_DIST_ONE_STREAM=1"; do
  timeout 300 env $env python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/gpu_dist_overlap.py 8192 512 3 2>&1 | grep "dist lu" | tee -a $O
done
rm -rf gpurun_out/trace_dist
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_dist -o d -- python tools/gpu_dist_overlap.py 8192 512 1 > gpurun_out/trace_dist.log 2>&1
f=$(find gpurun_out/trace_dist -name "*kernel_trace.csv" | head -1)
TL_WHOLE=2 python tools/trace_timeline.py $f >> $O 2>&1
rm -rf gpurun_out/trace_dist
tail -40 $O | cut -c1-150
