"""LU residual / checksum for a list of sizes (A/B of library builds through FAER_HIP_LIB).  GPU box only."""
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
dt = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else torch.float64
for n in [int(v) for v in sys.argv[2:]]:
    g = torch.Generator(device="cuda").manual_seed(n)
    a = torch.randn((n, n), dtype=dt, device="cuda", generator=g).t()
    work = a.clone()
    perm = F.partial_piv_lu_factor_in_place(work)[0]
    p = torch.as_tensor(perm.astype(np.int64), device="cuda")
    L = torch.tril(work, -1) + torch.eye(n, dtype=dt, device="cuda")
    r = (L @ torch.triu(work) - a[p]).abs().max().item()
    print(f"n={n} residual {r:.3e} checksum {float(work.double().sum().item())!r} permsum {int((perm.astype(np.int64) * np.arange(n)).sum())}", flush=True)
