#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c8
timeout 300 env FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_leaf_phases.py 2>&1 | grep -v amdgpu | grep -A1 "n=128 k=8192\|n=128 k=64 " | tee ${O}_leaf_phases.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "(test_trsm or plu or llt_vs or llt_ill or lu_solve_ill or qr_solve_ill or singular_diag or test_llt_solve or ldlt or test_qr_full_rank) and not 2000" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -8 ${O}_pytest.log
for env in "X=1" "FAER_HIP_LU_LEFT_DEFER=1" "FAER_HIP_LU_POLL_DELAY=4" "FAER_HIP_LU_POLL_DELAY=10" "FAER_HIP_LU_PANEL=2"; do
  timeout 200 env $env python tools/gpu_exp_one.py lu 16384 2>&1 | grep -v amdgpu
done
timeout 200 python tools/gpu_exp_one.py llt 16384 2>&1 | grep -v amdgpu
for wl in llt lu; do
  rm -rf gpurun_out/prof_r2c8_$wl
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2c8_$wl -o $wl -- python bench.py --workload $wl --steps 5 --warmup 2 --no-extras --no-cpu > ${O}_prof_$wl.log 2>&1; echo "prof $wl rc=$?"
  grep -o '"ms_per_step": [0-9.]*' ${O}_prof_$wl.log
done
timeout 200 python bench.py --workload qr --steps 5 --warmup 2 --no-extras --no-cpu 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
