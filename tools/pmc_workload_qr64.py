"""Workload for the HBM-traffic PMC passes of the one-pass tall QR (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs):
two factorizations of the bench shape (5e5 x 256 fp64) and a torch device copy of the same 1.02 GB as the yardstick
(wide vector loads: FETCH_SIZE reports half of the bytes read, MI355X_MICROARCH.md section HBM)."""
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
m, n = 500000, 256
g = torch.Generator(device="cuda").manual_seed(5)
a = torch.randn((n, m), dtype=torch.float64, device="cuda", generator=g).t()
h = torch.zeros((n, F.qr_recommended_block_size(m, n, np.float64)), dtype=torch.float64, device="cuda").t()
for _ in range(2):
    w = a.clone()  # the yardstick: 1.02 GB read + 1.02 GB written by a copy kernel
    assert F.qr_factor_in_place(w, h) == n
F.synchronize()
torch.cuda.synchronize()
print("pmc qr workload done")
