"""Distributed Cholesky (device backend) with ONE rank on one GPU: ms per factorization; under `rocprofv3 --kernel-trace` the
per-stream timeline of the rank (TL_WHOLE=2 python tools/trace_timeline.py).  usage: gpu_dist_llt_one.py [n] [nb] [reps] [rccl]
(`rccl`: the library's own RCCL transport with a one-rank communicator instead of a no-op Python callback; GPU_MAX_HW_QUEUES=8 as in bench.py)"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
use_rccl = len(sys.argv) > 4 and sys.argv[4] == "rccl"
F = ge.load_package()
torch.cuda.set_device(0)
F.lib()
F.use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(3)
b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
a = (b @ b.t() / n + 2 * torch.eye(n, dtype=torch.float64, device="cuda")).t()
del b
def single():
    t = 1e9
    for rep in range(3):
        loc = a.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        F.llt_factor_in_place(loc)
        torch.cuda.synchronize()
        if rep > 0:
            t = min(t, time.perf_counter() - t0)
    return t * 1e3


notes = []
skip_single = bool(os.environ.get("DLLT_NO_SINGLE"))
if not skip_single:
    notes.append(f"single-GPU driver before: {single():.2f} ms")
tr = F.RcclTransport(F.RcclTransport.unique_id(), 0, 1) if use_rccl else None
if use_rccl and not skip_single:
    notes.append(f"single-GPU driver with the communicator open: {single():.2f} ms")
best = 1e9
for rep in range(reps + 1):
    loc = a.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if tr is not None:
        F.dist_llt(loc, n, nb, 0, 1, transport=tr)
    else:
        F.dist_llt(loc, n, nb, 0, 1, lambda t, root: None)
    torch.cuda.synchronize()
    if rep > 0 or reps == 0:
        best = min(best, time.perf_counter() - t0)
L = torch.tril(loc)
x = torch.randn((n, 2), dtype=torch.float64, device="cuda")
r = (L @ (L.t() @ x) - a @ x).abs().max().item()
print(f"dist llt n={n} nb={nb} world=1 ({'rccl transport' if use_rccl else 'no-op callback'}): {best * 1e3:.2f} ms, residual {r:.2e}; " + "; ".join(notes))
