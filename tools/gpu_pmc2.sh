#!/bin/bash
# PMC passes (each counter set in its own run, --kernel-trace only: MI355X_MICROARCH.md, rocprofv3 PMC slots)
#   bash tools/gpu_pmc2.sh [workload.py] [tag]     -> gpurun_out/<tag>_pmc_summary.json (tools/pmc_summarize.py)
wl=${1:-tools/pmc_workload.py}
tag=${2:-r02}
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  t=$(echo $ctr | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$t -o p -- python $wl > $out/$t.log 2>&1; echo "$t rc=$?"
done
python tools/pmc_summarize.py $out gpurun_out/${tag}_pmc_summary.json
