#!/bin/bash
# kernel trace of the one-rank distributed Cholesky: timeline summary + kernel rows of a window.  usage: gpu_dllt_trace.sh [nb] [from_us] [to_us]
export TMPDIR=/tmp
nb=${1:-512}; lo=${2:-3000}; hi=${3:-8000}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp; rm -rf /tmp/pl
DLLT_NO_SINGLE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl -o llt -- python $root/tools/gpu_dist_llt_one.py 16384 $nb 1 $4 > /tmp/pl.log 2>&1; tail -1 /tmp/pl.log
f=$(find /tmp/pl -name "*kernel_trace.csv" | head -1)
cd $root
TL_WHOLE=2 python tools/trace_timeline.py $f > gpurun_out/dllt_timeline.txt 2>&1
python - $f $lo $hi <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "fh::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]
t0=int(rows[0]["Start_Timestamp"])
lo,hi=float(sys.argv[2])*1e3,float(sys.argv[3])*1e3
out=open("gpurun_out/dllt_rows.txt","w")
for r in rows:
    s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
    if lo<s<hi:
        out.write("%9.1f %9.1f q%s %7.1f %s grid %s\n"%(s/1e3,e/1e3,r["Queue_Id"],(e-s)/1e3,r["Kernel_Name"].replace("void fh::","").split("(")[0][:50],r["Grid_Size_X"]))
PY
head -24 gpurun_out/dllt_timeline.txt
