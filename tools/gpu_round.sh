#!/bin/bash
# One visit to a GPU box: parity tests, the bench line, kernel-trace profiles (csv summaries) of every workload.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r01_final'
tag=${1:-run}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python tools/gpu_fuzz.py 300 7 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json
for wl in ${PROF_WORKLOADS:-gemm llt lu qr tridiag}; do
  rm -rf gpurun_out/prof_${tag}_$wl
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$wl -o $wl -- python bench.py --workload $wl --steps 10 --warmup 2 --no-extras --no-cpu > gpurun_out/prof_${tag}_$wl.log 2>&1; echo "prof $wl rc=$?"
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_${tag}_$wl.log
done
