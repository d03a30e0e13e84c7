#!/bin/bash
# per-step kernel durations of a level-2 factorization at N = 4096 (bench workload $1, kernel name substring $2) by step index
export TMPDIR=/tmp
wl=$1; kn=$2
bash tools/gpu_visit.sh l2s prof:$wl | head -6
f=$(find gpurun_out/prof_l2s_$wl -name "*kernel_trace.csv" | head -1)
python - $f $kn <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-4096:]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
for k in (0,256,512,768,1024,1536,2048,2560,3072,3584,3840,4000,4090):
    print(k, "dur %.1f us"%(sum(d[k:k+4])/4))
print("sum dur %.1f ms"%(sum(d)/1e3))
PY
