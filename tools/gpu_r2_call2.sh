#!/bin/bash
# round 2, visit 2: new tests, substitution-leaf latency, LU panel A/B (one hop vs two hops per column), profiles
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c2
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "stability or scatter_index or host_submatrix or norm_l2_scaling or clients_compute or plu or lu_ or llt or ldlt or two_ranks or trsm" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -25 ${O}_pytest.log
timeout 300 python tools/gpu_micro_trsm.py > ${O}_trsm.txt 2>&1; grep -v amdgpu ${O}_trsm.txt
: > ${O}_ab.log
for v in 2 3; do for n in 1024 4096 16384; do
  timeout 200 env FAER_HIP_LU_PANEL=$v python tools/gpu_exp_one.py lu $n >> ${O}_ab.log 2>&1 || echo "FAILED panel=$v n=$n" >> ${O}_ab.log
done; done
grep -v amdgpu ${O}_ab.log
for wl in llt lu; do
  rm -rf gpurun_out/prof_r2c2_$wl
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2c2_$wl -o $wl -- python bench.py --workload $wl --steps 5 --warmup 2 --no-extras --no-cpu > ${O}_prof_$wl.log 2>&1; echo "prof $wl rc=$?"
  grep -o '"ms_per_step": [0-9.]*' ${O}_prof_$wl.log
done
timeout 200 python bench.py --workload qr --steps 5 --warmup 2 --no-extras --no-cpu 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
