#!/bin/bash
export TMPDIR=/tmp
python - <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
F = ge.load_package(); F.lib(); torch.cuda.set_device(0); F.use_torch_stream()
for (m, n) in ((2048, 2048), (4096, 4096), (100000, 256), (16384, 1024)):
    a = torch.randn((n, m), dtype=torch.float64, device="cuda").t()
    bs = F.qr_recommended_block_size(m, n, np.float64)
    h = torch.zeros((min(m, n), bs), dtype=torch.float64, device="cuda").t()
    w = a.clone()
    F.colpiv_qr_factor_in_place(w, h)
    w.copy_(a); torch.cuda.synchronize(); t0 = time.perf_counter()
    F.colpiv_qr_factor_in_place(w, h); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = min(m, n)
    byts = sum(2.0 * (m - k) * (n - k) * 8 for k in range(s))
    print(f"colpiv_qr f64 {m}x{n}: {dt*1e3:.1f} ms, {byts/dt/1e12:.2f} TB/s algorithmic (one read + one write of the trailing matrix per step)")
PY
