"""Latency of the triangular solve (substitution leaf + MFMA products) per shape, both orientations.
usage: python tools/gpu_micro_trsm.py  -> lines "n k side  us"   (side L: T X = B, lanes through the LDS tile;
side R: X T^T = B on a column-major panel, lanes along the rows)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for dt in (torch.float64, torch.float32):
    for n, k in [(64, 64), (128, 64), (128, 128), (128, 512), (128, 8192), (128, 16384), (256, 256), (512, 64), (512, 512),
                 (1024, 15360), (2048, 2048), (8192, 8192)]:
        t = (torch.tril(torch.randn((n, n), dtype=dt, device="cuda")) / n + torch.eye(n, dtype=dt, device="cuda")).t().contiguous().t()
        xl = torch.randn((k, n), dtype=dt, device="cuda").t()  # n x k column major
        xr = torch.randn((n, k), dtype=dt, device="cuda")      # n x k with unit stride along k (a transposed panel)
        us_l = timeit(lambda: F.solve_lower_triangular_in_place(t, xl))
        us_r = timeit(lambda: F.solve_lower_triangular_in_place(t, xr))
        flops = n * n * k
        print(f"{str(dt)[6:]:8s} n={n:5d} k={k:6d}  left {us_l:9.1f} us ({flops / us_l * 1e-6:7.2f} TF/s)   right {us_r:9.1f} us ({flops / us_r * 1e-6:7.2f} TF/s)")
