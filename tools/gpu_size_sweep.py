"""Latency of the factorizations over a range of sizes (device-resident operands, one GPU)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()


def bench(fn, reset, reps):
    reset()
    fn()
    best = 1e9
    for _ in range(reps):
        reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


print("n       dgemm      llt       lu        qr   (ms, fp64, best of reps)")
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 128, 256, 512, 1024, 2048, 4096, 8192]
for n in sizes:
    reps = 20 if n <= 1024 else 5
    a = torch.randn((n, n), dtype=torch.float64, device="cuda").t()
    b = torch.randn((n, n), dtype=torch.float64, device="cuda").t()
    c = torch.empty((n, n), dtype=torch.float64, device="cuda").t()
    spd = (a @ a.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t().contiguous().t()
    w = a.clone()
    t_mm = bench(lambda: F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0), lambda: None, reps)
    t_llt = bench(lambda: F.llt_factor_in_place(w), lambda: w.copy_(spd), reps)
    t_lu = bench(lambda: F.partial_piv_lu_factor_in_place(w), lambda: w.copy_(a), reps)
    bs = F.qr_recommended_block_size(n, n, np.float64)
    h = torch.zeros((n, bs), dtype=torch.float64, device="cuda").t()
    t_qr = bench(lambda: F.qr_factor_in_place(w, h), lambda: w.copy_(a), reps)
    print(f"{n:5d} {t_mm:9.3f} {t_llt:9.3f} {t_lu:9.3f} {t_qr:9.3f}", flush=True)
