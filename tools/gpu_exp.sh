#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 200 python tools/gpu_stress_llt.py 2>&1 | grep -v amdgpu | tr '\n' ';'; echo; done
for i in 1 2 3; do timeout 200 python tools/gpu_stress_llt.py side 2>&1 | grep -v amdgpu | tr '\n' ';'; echo; done
for i in 1 2 3; do timeout 200 python tools/gpu_stress_lu.py 2>&1 | grep -v amdgpu | tr '\n' ';'; echo; done
