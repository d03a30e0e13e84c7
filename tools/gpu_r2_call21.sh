#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r2c21
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "tridiag or bidiag or hessenberg or qr or householder" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -4 ${O}_pytest.log | cut -c1-200
for wl in tridiag bidiag hess cpqr; do
  for env in "X=1" "FAER_HIP_TBLOCK_BATCHED=0"; do
    echo "$wl $env: $(timeout 300 env $env python bench.py --workload $wl --steps 3 --warmup 1 --no-extras --no-cpu 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  done
done
