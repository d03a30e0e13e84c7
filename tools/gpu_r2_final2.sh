#!/bin/bash
# final visit of round 2: full GPU suite + smoke + fuzz, then (the pool has a fast and a slow population of boxes) the bench
# line and kernel-trace summaries tagged by population
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python tools/gpu_fuzz.py 300 7 2>&1 | tail -1
probe=$(timeout 200 python tools/gpu_exp_one.py lu 16384 2>&1 | grep -o "lu n=16384: [0-9.]*" | grep -o "[0-9.]*$")
slow=$(python -c "print(1 if float('$probe' or 999) > 132 else 0)")
tag=r02g; [ "$slow" = "1" ] && tag=r02g_slow
echo "probe lu = $probe ms -> $tag"
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json
for wl in gemm llt lu qr tridiag bidiag hess; do
  rm -rf gpurun_out/prof_${tag}_$wl
  steps=10; [ "$wl" = "tridiag" -o "$wl" = "bidiag" -o "$wl" = "hess" ] && steps=3
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$wl -o $wl -- python bench.py --workload $wl --steps $steps --warmup 1 --no-extras --no-cpu > gpurun_out/prof_${tag}_$wl.log 2>&1; echo "prof $wl rc=$?"
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_${tag}_$wl.log
  f=$(find gpurun_out/prof_${tag}_$wl -name "*kernel_trace.csv" | head -1)
  if [ "$wl" = "lu" -o "$wl" = "llt" ]; then python tools/trace_timeline.py $f > gpurun_out/${tag}_timeline_$wl.txt 2>&1; fi
  rm -f $f
done
