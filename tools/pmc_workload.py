"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs):
  * the headline DGEMM (N = 8192, Replace), two launches;
  * a calibration launch of the SAME kernel with known traffic: M = N = 8192, K = 16, Accum::Add
    (reads C once = 512 MiB + 2 MiB of operands, writes C once = 512 MiB);
  * a torch device copy of 1 GiB (wide vector loads) as a second yardstick."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()


def cm(m, n):
    return torch.randn((n, m), dtype=torch.float64, device="cuda").t()


n = 8192
a, b, c = cm(n, n), cm(n, n), cm(n, n)
for _ in range(2):
    F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0)
a16, b16 = cm(n, 16), cm(16, n)
for _ in range(2):
    F.matmul(c, F.ACCUM_ADD, a16, b16, 1.0)
# round 2 additions (north_star: "achieved fp64 MFMA utilisation on the trailing-matrix GEMM update, achieved HBM GB/s on
# the skinny panel step"): the SYRK-shaped trailing update of the Cholesky (r = 15360, K = 1024, lower), the trailing
# GEMM of the LU (15872 x 15872 x 512), one LU (N = 4096) and one QR (2e5 x 64 fp32) for their panel kernels
r, k = 15360, 1024
xp, cs = cm(r, k), cm(r, r)
F.gemm(cs, F.DST_LOWER, F.ACCUM_ADD, xp, xp.t(), -1.0)
lp, up, tr = cm(15872, 512), cm(512, 15872), cm(15872, 15872)
F.matmul(tr, F.ACCUM_ADD, lp, up, -1.0)
del cs, tr, xp, lp, up
torch.cuda.empty_cache()
al = cm(4096, 4096)
F.partial_piv_lu_factor_in_place(al)
import numpy as np  # noqa: E402

aq = torch.randn((64, 200000), dtype=torch.float32, device="cuda").t()
hq = torch.zeros((64, F.qr_recommended_block_size(200000, 64, np.float32)), dtype=torch.float32, device="cuda").t()
F.qr_factor_in_place(aq, hq)
x = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
y.copy_(x)
F.synchronize()
torch.cuda.synchronize()
print("pmc workload done")
