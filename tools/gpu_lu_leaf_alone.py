"""LU of tall m x w panels on an otherwise idle chip: what a 64-column leaf / a flat 512-column panel cost without neighbours
(the in-situ figures come from the kernel traces of the full factorization).  GPU box only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
for m, w in ((2048, 64), (4096, 64), (8192, 64), (16384, 64), (8192, 256), (8192, 512), (16384, 512)):
    g = torch.Generator(device="cuda").manual_seed(4)
    a = torch.randn((w, m), dtype=torch.float64, device="cuda", generator=g).t()
    work = a.clone()
    best = 1e9
    for _ in range(6):
        work.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        F.partial_piv_lu_factor_in_place(work)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"lu {m} x {w}: {best * 1e3:.1f} us = {best * 1e3 / w:.2f} us per column (call incl. host sync)", flush=True)
