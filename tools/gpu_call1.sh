#!/bin/bash
# one GPU-box visit: parity tests, bench line, kernel-trace profile (csv summaries)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
cat gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
rm -rf gpurun_out/prof_all
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_all -o all -- python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/prof_all.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_all -name '*stats*' | head
