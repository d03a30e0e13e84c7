#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$ctr
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -o p -- python tools/pmc_workload.py > gpurun_out/pmc_$ctr.log 2>&1; echo "$ctr rc=$?"
done
ls gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
cat gpurun_out/bench_default.json
timeout 300 python -m pytest tests/test_gpu_factor.py -m gpu -q -x -k dist 2>&1 | tail -3
