"""Distributed Cholesky (device backend) with ONE rank on one GPU: ms per factorization; under `rocprofv3 --kernel-trace` the
per-stream timeline of the rank (TL_WHOLE=2 python tools/trace_timeline.py).  usage: gpu_dist_llt_one.py [n] [nb] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
F = ge.load_package()
torch.cuda.set_device(0)
F.lib()
F.use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(3)
b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
a = (b @ b.t() / n + 2 * torch.eye(n, dtype=torch.float64, device="cuda")).t()
del b
best = 1e9
for rep in range(reps + 1):
    loc = a.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    F.dist_llt(loc, n, nb, 0, 1, lambda t, root: None)
    torch.cuda.synchronize()
    if rep > 0 or reps == 0:
        best = min(best, time.perf_counter() - t0)
L = torch.tril(loc)
x = torch.randn((n, 2), dtype=torch.float64, device="cuda")
r = (L @ (L.t() @ x) - a @ x).abs().max().item()
print(f"dist llt n={n} nb={nb} world=1: {best * 1e3:.2f} ms, residual {r:.2e}")
