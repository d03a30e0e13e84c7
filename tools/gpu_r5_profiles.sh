#!/bin/bash
# Round-5 evidence in one visit: kernel traces (stats + timelines) of the four workloads, PMC passes (GEMM / LLT / LU updates, QR)
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_r5_visit.sh r05 prof:gemm prof:llt prof:lu prof:qr > gpurun_out/r05_prof.log 2>&1
bash tools/gpu_pmc2.sh tools/pmc_workload.py r05 > gpurun_out/r05_pmc.log 2>&1
bash tools/gpu_pmc2.sh tools/pmc_workload_qr.py r05qr > gpurun_out/r05_pmcqr.log 2>&1
rm -rf gpurun_out/pmc_r05 gpurun_out/pmc_r05qr
ls -la gpurun_out | grep r05
