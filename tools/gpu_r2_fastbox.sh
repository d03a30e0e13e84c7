#!/bin/bash
# the pool has two populations of boxes (latency-bound kernels 30-50 % apart): probe first, record the bench line and the
# kernel-trace summaries only on a box of the fast population (tag r02f); a slow box costs ~1 minute
export TMPDIR=/tmp
mkdir -p gpurun_out
probe=$(timeout 200 python tools/gpu_exp_one.py lu 16384 2>&1 | grep -o "lu n=16384: [0-9.]*" | grep -o "[0-9.]*$")
echo "probe lu = $probe ms"
slow=$(python -c "print(1 if float('$probe' or 999) > 130 else 0)")
if [ "$slow" = "1" ]; then echo "SLOW BOX"; exit 0; fi
tag=r02h
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json
for wl in llt lu qr; do
  rm -rf gpurun_out/prof_${tag}_$wl
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$wl -o $wl -- python bench.py --workload $wl --steps 10 --warmup 2 --no-extras --no-cpu > gpurun_out/prof_${tag}_$wl.log 2>&1; echo "prof $wl rc=$?"
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_${tag}_$wl.log
  f=$(find gpurun_out/prof_${tag}_$wl -name "*kernel_trace.csv" | head -1)
  python tools/trace_timeline.py $f > gpurun_out/${tag}_timeline_$wl.txt 2>&1
  rm -f $f
done
