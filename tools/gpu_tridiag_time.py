"""timing of faer_hip_tridiag_in_place (device operands), algorithmic bytes: sum_k (n-k-2)^2 / 2 entries read + written"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gpu_util import init_gpu, to_dev
import torch

F = init_gpu()
for dtype in (np.float64, np.float32):
    for n in (1024, 4096):
        rng = np.random.default_rng(1)
        a = rng.standard_normal((n, n)); a = np.asarray(a + a.T, dtype=dtype, order="F")
        best = 1e9
        for rep in range(3):
            vd, hd = to_dev(a), to_dev(np.zeros((32, n - 1), dtype=dtype, order="F"))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            F.tridiag_in_place(vd, hd)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        r = np.arange(n - 2, 0, -1, dtype=np.float64)
        byts = (r * r / 2).sum() * 2 * np.dtype(dtype).itemsize
        print(f"tridiag {np.dtype(dtype).name} n={n}: {best*1e3:.2f} ms, {byts/best/1e9:.0f} GB/s algorithmic, {4*n**3/3/best/1e9:.0f} GFLOP/s")
