#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c9
timeout 300 env FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_leaf_phases.py 2>&1 | grep -v amdgpu | grep -A1 "float64 n=128 k=8192" | tee ${O}_leaf_phases.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "(test_trsm or plu or llt_vs or llt_ill or lu_solve_ill or singular_diag or test_llt_solve or ldlt_vs or test_qr_full_rank or rccl) and not 2000" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 ${O}_pytest.log
for env in "X=1" "FAER_HIP_LLT_TAIL=2048" "FAER_HIP_LLT_TAIL=3072" "FAER_HIP_LLT_TAIL=6144" "FAER_HIP_LLT_DPANEL=4096" "FAER_HIP_PANEL_CUS=48"; do
  timeout 200 env $env python tools/gpu_exp_one.py llt 16384 2>&1 | grep -v amdgpu
done
for env in "X=1" "FAER_HIP_LU_PANEL=3" "FAER_HIP_PANEL_CUS=48" "FAER_HIP_PANEL_CUS=64"; do
  timeout 200 env $env python tools/gpu_exp_one.py lu 16384 2>&1 | grep -v amdgpu
done
timeout 200 python bench.py --workload qr --steps 5 --warmup 2 --no-extras --no-cpu 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
