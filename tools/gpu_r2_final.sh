#!/bin/bash
# round 2: full GPU suite, smoke, fuzz, bench line, kernel-trace profiles of every workload, PMC passes
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_round.sh ${1:-r02_final}
bash tools/gpu_pmc2.sh
timeout 120 env FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_leaf_phases.py 2>&1 | grep -v amdgpu > gpurun_out/r02_final_leaf_phases.txt
