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
x = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
y.copy_(x)
F.synchronize()
torch.cuda.synchronize()
print("pmc workload done")
