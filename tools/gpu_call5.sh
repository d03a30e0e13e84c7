#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/gpu_micro.py > gpurun_out/micro.log 2>&1; echo "micro rc=$?"
cat gpurun_out/micro.log
