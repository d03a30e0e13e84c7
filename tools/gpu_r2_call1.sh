#!/bin/bash
# round 2, visit 1: parity of the substitution TRSM + its effect on the factorizations
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x -k "stability or trsm" > gpurun_out/r2c1_pytest_trsm.log 2>&1; echo "pytest trsm rc=$?"
tail -30 gpurun_out/r2c1_pytest_trsm.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r2c1_pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -40 gpurun_out/r2c1_pytest_all.log
for wl in llt lu qr; do
  timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-extras --no-cpu > gpurun_out/r2c1_bench_$wl.json 2> gpurun_out/r2c1_bench_$wl.err; echo "bench $wl rc=$?"
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2c1_bench_$wl.json
done
timeout 300 python tools/gpu_size_sweep.py > gpurun_out/r2c1_sweep.txt 2>&1; tail -12 gpurun_out/r2c1_sweep.txt
