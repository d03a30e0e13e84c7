#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_factor.py -m gpu -q -x -k "full_piv" 2>&1 | tail -15
python - <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
F = ge.load_package(); F.lib(); torch.cuda.set_device(0); F.use_torch_stream()
for n in (2048, 4096, 8192):
    a = torch.randn((n, n), dtype=torch.float64, device="cuda").t()
    w = a.clone()
    F.full_piv_lu_factor_in_place(w)
    w.copy_(a); torch.cuda.synchronize(); t0 = time.perf_counter()
    F.full_piv_lu_factor_in_place(w); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    byts = sum(2 * (n - k) ** 2 * 8 for k in range(1, n))
    print(f"full_piv_lu f64 n={n}: {dt*1e3:.1f} ms, {2*n**3/3/dt/1e9:.1f} GFLOP/s, {byts/dt/1e12:.2f} TB/s algorithmic")
PY
