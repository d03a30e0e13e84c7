#!/bin/bash
# host API timeline between two pack launches of the one-rank distributed Cholesky (RCCL transport)
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; rm -rf /tmp/ph
DLLT_NO_SINGLE=1 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/ph -o llt -- python $root/tools/gpu_dist_llt_one.py 16384 512 1 rccl > /tmp/ph.log 2>&1; tail -1 /tmp/ph.log
ls /tmp/ph/*/ 2>/dev/null | head
k=$(find /tmp/ph -name "*kernel_trace.csv" | head -1); h=$(find /tmp/ph -name "*hip_api_trace.csv" | head -1)
python - $k $h <<PY
import csv,sys
ks=[r for r in csv.DictReader(open(sys.argv[1]))]
ks.sort(key=lambda r:int(r["Start_Timestamp"]))
cp=[r for r in ks if "copy_kernel" in r["Kernel_Name"]]
hs=[r for r in csv.DictReader(open(sys.argv[2]))]
print(list(hs[0].keys()))
hs.sort(key=lambda r:int(r["Start_Timestamp"]))
# a pair of copies late in the run
n=len(cp); a,b=cp[int(n*0.8)],cp[int(n*0.8)+1]
t0=int(a["Start_Timestamp"]); t1=int(b["Start_Timestamp"])
print("copies at", 0, (t1-t0)/1e3, "us apart; kernel a dur", (int(a["End_Timestamp"])-t0)/1e3)
for r in hs:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if t0-400e3 < s < t1+50e3:
        print("%9.1f %7.1f %s"%((s-t0)/1e3,(e-s)/1e3,r["Function"]))
PY
