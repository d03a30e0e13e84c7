"""Distributed LU (device backend) on ONE GPU: how much of the owner's panel factorization overlaps with the trailing
updates inside a rank.  One process (world size 1: every block column is owned, no transport) or several processes
sharing the GPU (gloo on host copies, launched by torch.distributed.run).  Prints ms per factorization; under
`rocprofv3 --kernel-trace` the one-process run gives the per-stream timeline (tools/trace_timeline.py).
usage: gpu_dist_overlap.py [n] [nb] [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
F = ge.load_package()
torch.cuda.set_device(0)
F.lib()
F.use_torch_stream()
if world > 1:
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
g = torch.Generator(device="cuda").manual_seed(1234)
a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
cols = [c for b in range(rank, (n + nb - 1) // nb, world) for c in range(b * nb, min(n, (b + 1) * nb))]
loc0 = a[:, cols].t().contiguous().t()


def bcast(t, root):
    if world == 1:
        return
    torch.cuda.current_stream().synchronize()
    h = t.cpu()
    dist.broadcast(h, src=root)
    t.copy_(h)


best = 1e9
for rep in range(reps + 1):  # (reps = 0: a single cold run, for traces)
    loc = loc0.clone()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    F.dist_partial_piv_lu(loc, n, nb, rank, world, bcast)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if rep > 0 or reps == 0:
        best = min(best, time.perf_counter() - t0)
if rank == 0:
    mode = "one stream" if os.environ.get("FAER_HIP_DIST_ONE_STREAM") else "bulk + panel streams"
    print(f"dist lu n={n} nb={nb} world={world} ({mode}): {best * 1e3:.2f} ms")
