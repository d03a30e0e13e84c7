#!/bin/bash
# round 2 PMC passes (each counter set in its own run, --kernel-trace only: MI355X_MICROARCH.md, rocprofv3 PMC slots)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  tag=$(echo $ctr | tr ' ' '_')
  rm -rf gpurun_out/pmc2/$tag
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc2/$tag -o p -- python tools/pmc_workload.py > gpurun_out/pmc2_$tag.log 2>&1; echo "$tag rc=$?"
done
python tools/pmc_summarize.py gpurun_out/pmc2 gpurun_out/r02_pmc_summary.json
