#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -x -k "dist or rccl" > gpurun_out/r2c24_pytest.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r2c24_pytest.log
bash tools/gpu_dist_overlap.sh > /dev/null 2>&1
head -30 gpurun_out/r02_dist_overlap.txt | cut -c1-140
