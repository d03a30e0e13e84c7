"""Dump the LU factors of one size (debugging aid; A/B through FAER_HIP_LIB / env).  GPU box only."""
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
n, out = int(sys.argv[1]), sys.argv[2]
g = torch.Generator(device="cuda").manual_seed(n)
a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
work = a.clone()
perm = F.partial_piv_lu_factor_in_place(work)[0]
np.savez(out, lu=work.cpu().numpy(), perm=perm.astype(np.int64))
