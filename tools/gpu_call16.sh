#!/bin/bash
export TMPDIR=/tmp
timeout 400 python tools/gpu_stress_llt.py 2>&1 | grep -v amdgpu
timeout 400 python tools/gpu_stress_llt.py side 2>&1 | grep -v amdgpu
