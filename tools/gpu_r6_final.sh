#!/bin/bash
# Final visit of round 6: default bench line, one-rank dry runs of the distributed drivers, kernel summaries and timelines of the
# workloads that changed at the end of the round.   gpurun --timeout 1500 -- 'bash tools/gpu_r6_final.sh'
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err; echo "bench rc=$?"
BENCH_FORCE_DIST=1 timeout 400 python bench.py --no-cpu > gpurun_out/r06f_dist_lu.json 2> gpurun_out/r06f_dist_lu.err; echo "dist lu rc=$?"
BENCH_FORCE_DIST=1 timeout 400 python bench.py --workload llt --no-cpu > gpurun_out/r06f_dist_llt.json 2> gpurun_out/r06f_dist_llt.err; echo "dist llt rc=$?"
for f in r06f_bench r06f_dist_lu r06f_dist_llt; do python - gpurun_out/$f.json <<PY
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o=d.get("others",{})
print(sys.argv[1], d["config"]["workload"], d["ms_per_step"], d.get("speedup_vs_1gpu_same_run"), {k:(v.get("ms") if isinstance(v,dict) else v) for k,v in o.items()})
PY
done
bash tools/gpu_visit.sh r06f prof:llt prof:fplu qrprof > gpurun_out/r06f_prof.log 2>&1; grep -E "rc=|ms_per_step" gpurun_out/r06f_prof.log
f=$(find gpurun_out/prof_r06f_llt -name "*kernel_trace.csv" | head -1); python tools/trace_timeline.py $f > gpurun_out/r06f_llt_timeline.txt 2>&1
f=$(find gpurun_out/prof_r06f_qr -name "*kernel_trace.csv" | head -1); python tools/qr_tail_timeline.py $f > gpurun_out/r06f_qr_timeline.txt 2>&1; tail -1 gpurun_out/r06f_qr_timeline.txt
bash tools/gpu_dllt_trace.sh 512 40000 46000 rccl > gpurun_out/r06f_dllt_trace.log 2>&1; cp gpurun_out/dllt_timeline.txt gpurun_out/r06f_dist_llt_timeline.txt; head -3 gpurun_out/dllt_timeline.txt
