#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== default"; timeout 300 python bench.py --no-cpu --steps 3 --warmup 1 2>&1 | grep -o '"others".*' | cut -c1-330
echo "== default again"; timeout 300 python bench.py --no-cpu --steps 3 --warmup 1 2>&1 | grep -o '"others".*' | cut -c1-330
echo "== no lookahead"; FAER_HIP_NO_LOOKAHEAD=1 timeout 300 python bench.py --no-cpu --steps 3 --warmup 1 2>&1 | grep -o '"others".*' | cut -c1-330
echo "== llt direct"; timeout 300 python bench.py --workload llt --no-cpu --no-extras --steps 3 --warmup 1 2>&1 | grep -o '"value": [0-9.]*'
