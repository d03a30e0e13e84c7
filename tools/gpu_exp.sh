#!/bin/bash
# Template of a one-visit A/B experiment (edit, then: gpurun --timeout 900 -- 'bash tools/gpu_exp.sh').
# Everything compared must run inside ONE visit: the boxes of the pool differ by up to 15 % on latency-bound work.
#   * env switches of the library ("Environment switches") select code paths per process;
#   * FAER_HIP_LIB=/path/to/other/libfaer_hip.so loads another build (e.g. `git archive <commit> | tar -x -C /tmp/x &&
#     make -C /tmp/x/faer-rs_amd/csrc && cp .../libfaer_hip.so tools/ab/old.so`: untracked *.so files travel with gpurun).
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/exp.log
: > $O
run() { timeout 240 env "$@" python tools/gpu_exp_one.py $WHAT $N >> $O 2>&1 || echo "FAILED: $* $WHAT" >> $O; }
N=16384
WHAT=lu
run FAER_HIP_LU_SPLIT=0
run FAER_HIP_LU_SPLIT=1
WHAT=llt
run FAER_HIP_LLT_TAIL=0
run FAER_HIP_LLT_TAIL=4096
grep -v amdgpu.ids $O
