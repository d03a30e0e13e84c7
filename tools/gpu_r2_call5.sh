#!/bin/bash
# round 2, visit 5: pipelined substitution leaf, LU panel phase accounting, profiles
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c5
timeout 600 python -m pytest tests -m gpu -q --tb=short -x -k "(test_trsm or plu or llt_vs or llt_ill or lu_solve_ill or qr_solve_ill or singular_diag or test_llt_solve or ldlt_vs or two_ranks) and not 2000" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -6 ${O}_pytest.log
: > ${O}_phases.log
for n in 16384; do
  timeout 200 env FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_exp_one.py lu $n >> ${O}_phases.log 2>&1
done
timeout 200 env FAER_HIP_NO_LOOKAHEAD=1 FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_exp_one.py lu 16384 >> ${O}_phases.log 2>&1
grep -v amdgpu ${O}_phases.log | cut -c1-900
for v in 2 3; do timeout 200 env FAER_HIP_LU_PANEL=$v python tools/gpu_exp_one.py lu 16384 2>&1 | grep -v amdgpu; done
for wl in llt lu; do
  rm -rf gpurun_out/prof_r2c5_$wl
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2c5_$wl -o $wl -- python bench.py --workload $wl --steps 5 --warmup 2 --no-extras --no-cpu > ${O}_prof_$wl.log 2>&1; echo "prof $wl rc=$?"
  grep -o '"ms_per_step": [0-9.]*' ${O}_prof_$wl.log
done
timeout 200 python tools/gpu_exp_one.py llt 16384 2>&1 | grep -v amdgpu
timeout 200 python bench.py --workload qr --steps 5 --warmup 2 --no-extras --no-cpu 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
